#!/usr/bin/env python3
"""Development aid: stage-kernel time of the default benchmark block (TGV 256^3) for several values of the tuning
variables the library reads from the environment at every launch.

    python profiles/sweep.py VAR=v1,v2,... [VAR2=...] [scheme=senseur] [n=256]
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
from fast_amd import cases, FastC
import fast_amd.PyTree as FastS

args = dict(a.split('=') for a in sys.argv[1:])
scheme = args.pop('scheme', 'senseur')
n = int(args.pop('n', '256'))
t = cases.tgv(n, scheme)
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
names = list(args)
it = 1
for combo in itertools.product(*[args[k].split(',') for k in names]):
    for k, v in zip(names, combo):
        os.environ[k] = v
    ctx.compute(it, 1, 3, 1); it += 1
    ctx.stage_kernel_ms(reset=True)
    for _ in range(5):
        ctx.compute(it, 1, 3, 1); it += 1
    ms, nl = ctx.stage_kernel_ms(reset=True)
    print(' '.join('%s=%s' % kv for kv in zip(names, combo)), ' stage kernel %.4f ms (%d launches)' % (ms, nl), flush=True)
