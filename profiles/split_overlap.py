#!/usr/bin/env python3
"""Development aid: cost of the split stage (tiles that feed halo messages first, the messages while the rest of the
stage runs) on one GPU, with the library's RCCL transport sending the periodic faces of one block from the rank to
itself.  Prints ms per step of a 256^3 block without messages (BCs), and with messages for several k-chunk sizes.

    python profiles/split_overlap.py [cells]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
from fast_amd import cases, FastC
import fast_amd.PyTree as FastS
import fast_amd.Mpi as FastSmpi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def run(step, ctx, nsteps=40, warm=5):
    for it in range(1, warm + 1):
        step(it)
    ctx.sync()
    t0 = time.perf_counter()
    for it in range(warm + 1, warm + nsteps + 1):
        step(it)
    ctx.sync()
    return (time.perf_counter() - t0) / nsteps * 1e3


t = cases.tgv(n, 'senseur', time_step=0.003 * 32. / n)
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
print("no messages (periodic BCs): %.3f ms per step" % run(lambda it: ctx.compute(it, 1, 3, 1), ctx), flush=True)
ctx.close(); FastC.HOOK = None
for kc in sys.argv[2:] or ['0', '16', '32', '64']:
    if kc != '0':
        os.environ['FASTS_HIP_KCHUNK'] = kc
    else:
        os.environ.pop('FASTS_HIP_KCHUNK', None)
    for overlap in ('1', '0'):
        os.environ['FASTS_HIP_OVERLAP'] = overlap
        job = FastSmpi.tgv_job(n, (1, 1, 1), 'senseur', 'NSLaminar', 0, 1, 0, loopback=True)
        ms = run(job.step, job.ctx)
        z = job.ctx
        print("messages to itself, kchunk %s, overlap %s: %.3f ms per step" % (kc if kc != '0' else 'default', overlap, ms), flush=True)
        job.ctx.close(); FastC.HOOK = None
