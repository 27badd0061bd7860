#!/usr/bin/env python3
"""Development aid: time of the Cartesian stage kernel with one leg switched off at a time (library built with
-DFS_X_LEGS, tools/build_variant.sh legs -DFS_X_LEGS; the results of such runs are wrong, only the times count).

    FASTS_HIP_LIB=fast_amd/csrc/libfasts_hip_legs.so python profiles/legs.py [scheme] [n]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
from fast_amd import cases, FastC
import fast_amd.PyTree as FastS

scheme = sys.argv[1] if len(sys.argv) > 1 else 'senseur'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
t = cases.tgv(n, scheme)
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
for it in range(1, 4):
    ctx.compute(it, 1, 3, 1)
ctx.sync()
NAMES = {1: 'no stores', 2: 'no cell loads', 4: 'no plane DMA', 8: 'no arithmetic'}
print("%-44s %8s %8s %8s" % ("legs off", "stage 1", "stage 2", "stage 3"))
LIST = [int(v) for v in os.environ.get('FASTS_LEGS_LIST', '0,1,2,4,8,3,7,9,10,12,11,14,6,0').split(',')]
for x in LIST:
    os.environ['FASTS_HIP_XLEGS'] = str(x)
    row = []
    for s in (1, 2, 3):
        ctx.compute(1, s, s, 1)
        ctx.stage_kernel_ms(reset=True)
        for it in range(20):
            ctx.compute(2, s, s, 1)
        ctx.sync()
        ms, cnt = ctx.stage_kernel_ms(reset=False)
        row.append(ms)
    label = ' + '.join(NAMES[b] for b in (1, 2, 4, 8) if x & b) or 'full kernel'
    print("%-44s %8.4f %8.4f %8.4f" % (label, row[0], row[1], row[2]), flush=True)
