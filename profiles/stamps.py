#!/usr/bin/env python3
"""Development aid: phase durations inside one workgroup of the DMA stage kernel, from shader-clock
stamps (library built with -DFS_STAMP, see fasts_stage_dma.hip).

    FASTS_HIP_LIB=fast_amd/csrc/libfasts_hip_stamp.so FASTS_HIP_STAGE=dma python profiles/stamps.py [scheme]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_amd import cases, FastC, _capi
import fast_amd.PyTree as FastS

scheme = sys.argv[1] if len(sys.argv) > 1 else 'senseur'
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
t = cases.tgv(256, scheme)
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
for it in range(1, 4):
    ctx.compute(it, 1, 3, 1)
ctx.sync()
NW = 12
buf = np.zeros(NW * 40 * 8, dtype=np.int64)
rc = _capi.lib().fs_stamp_fetch(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
s = buf.reshape(NW, 40, 8)
print("last launch = stage 3; shader cycles per phase, steps 2..30 averaged, per wave of workgroup 0")
for w in range(NW):
    d = []
    for it in range(2, 30):
        row = s[w, it]
        nxt = s[w, it + 1, 0]
        if w < 8:
            ph = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4],
                  row[6] - row[5], row[7] - row[6], nxt - row[7]]
        else:
            ph = [row[1] - row[0], 0, 0, 0, row[5] - row[1], row[6] - row[5], row[7] - row[6], nxt - row[7]]
        d.append(ph)
    m = np.array(d, dtype=float).mean(axis=0)
    names = ['top', 'own+kface', 'finish+req', 'jface', 'iface', 'bar1', 'combine', 'bar2'] if w < 8 else \
            ['staging', '-', '-', '-', 'edge face', 'bar1', '-', 'bar2']
    print("wave %2d: total %7.0f | " % (w, m.sum()) + " | ".join("%s %6.0f" % (n, v) for n, v in zip(names, m)))
w0 = s[0, 2:30]
print("step period (wave 0):", np.diff(w0[:, 0]).mean())
