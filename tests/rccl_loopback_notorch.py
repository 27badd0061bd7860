"""Started by test_gpu_comm.py in a process of its own that never imports torch: the library then opens the system's
librccl (dlopen) next to the system HIP runtime, as bench.py does on a multi-GPU node.  One rank, joins as messages to
itself, compared with the local-join run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
from fast_amd import FastC, Internal
import fast_amd.Mpi as FastSmpi

assert 'torch' not in sys.modules


def run(loopback):
    job = FastSmpi.tgv_job(10, (1, 1, 2), 'senseur', 'NSLaminar', 0, 1, 0, loopback=loopback)
    for it in range(1, 4):
        FastSmpi._compute(job.t, job.metrics, it, job.tc, job.graph)
    out = [FastC.level_buffer(z).copy() for z in Internal.getZones(job.t)]
    s = FastSmpi.allreduce([1., 2.], 'sum')
    FastSmpi.barrier()
    job.ctx.close()
    return out, s


a, _ = run(False)
b, s = run(True)
ok = all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(s, [1., 2.]) and 'torch' not in sys.modules
print("RCCL_NOTORCH_RESULT %d" % int(ok), flush=True)
sys.exit(0 if ok else 1)
