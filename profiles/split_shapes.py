import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
from fast_amd import FastC
import fast_amd.Mpi as FastSmpi
def run(step, ctx, nsteps=30, warm=5):
    for it in range(1, warm + 1): step(it)
    ctx.sync(); t0 = time.perf_counter()
    for it in range(warm + 1, warm + nsteps + 1): step(it)
    ctx.sync(); return (time.perf_counter() - t0) / nsteps * 1e3
# the 8-GPU decomposition on one GPU, an eighth of the size: 8 blocks of 256 x 128 x 64 cells
for loop in (False, True):
    job = FastSmpi.tgv_job((256, 128, 64), (1, 2, 4), 'senseur', 'NSLaminar', 0, 1, 0, loopback=loop)
    print("8 blocks of 256x128x64 on one GPU, joins as %s: %.3f ms per step" % ("messages to itself" if loop else "local copies", run(job.step, job.ctx)), flush=True)
    job.ctx.close(); FastC.HOOK = None
# one block per GPU of the 8-GPU job: 512 x 256 x 128 cells, j and k faces as messages (here: to itself), i periodic
for loop in (False, True):
    job = FastSmpi.tgv_job((512, 256, 128), (1, 1, 1), 'senseur', 'NSLaminar', 0, 1, 0, loopback=loop)
    print("one block of 512x256x128, all faces as %s: %.3f ms per step" % ("messages to itself" if loop else "BCs", run(job.step, job.ctx)), flush=True)
    job.ctx.close(); FastC.HOOK = None
