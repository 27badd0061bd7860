"""One rank of the multi-rank GPU test (started by test_gpu_mpi.py): the product path
fast_amd.Mpi (warmup / _compute) with every rank on GPU 0; RCCL wants one GPU per rank, so the
messages travel over gloo, staged through the host, through the transport hook of warmup and
the split native calls.  The blocks are dealt to the ranks in consecutive runs (several blocks
per rank: local joins next to messages).  Each rank compares its blocks with a serial run of
the whole job by the CPU restatement.

    python mpi_gpu_worker.py <bi> <bj> <bk> <n> <nsteps> [explicit|implicit|implicit_threads|curvilinear|sa|walls|walls_implicit]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.distributed as dist

from fast_amd import cases, connect, FastC, Internal
import fast_amd.Mpi as FastSmpi
from oracle import oracle as O
from common import prepare, zone_arrays, wall_distance, rel_err
from halo_gloo import GlooTransport

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def main():
    blocks = tuple(int(a) for a in sys.argv[1:4])
    n = int(sys.argv[4]); nsteps = int(sys.argv[5])
    mode = sys.argv[6] if len(sys.argv) > 6 else 'explicit'
    nn = tuple(n * b for b in blocks)
    opts = {'time_step': 0.003 * 32. / max(nn)}
    threads = 1
    if mode in ('implicit', 'implicit_threads'):   # BDF + LU-SGS, the Newton iterates cross the rank boundary every sub-iteration
        opts = {'time_step': 0.02, 'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': 3}}
        if mode == 'implicit_threads':             # LU-SGS solve of every block in the blocks of 4 threads of the reference
            threads = 4
            FastC.OMP_NUM_THREADS = threads
    elif mode == 'curvilinear':
        opts['perturb'] = 0.15
    elif mode.startswith('walls'):   # curvilinear blocks between two walls: the i direction must not be split
        opts['perturb'] = 0.15
        opts['bcs'] = {'imin': 'BCWall', 'imax': 'BCFarfield'}
        if mode == 'walls_implicit':
            opts.update({'time_step': 0.01, 'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': 3}})
    model = 'NSTurbulent' if mode == 'sa' else 'NSLaminar'    # 6 variables per cell in the messages
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    job = FastSmpi.tgv_job(n, blocks, 'senseur', model, rank, world, 0, transport=GlooTransport(0), **opts)
    for it in range(1, nsteps + 1):
        FastSmpi._compute(job.t, job.metrics, it, job.tc, job.graph)
    got = [FastC.level_buffer(z) for z in Internal.getZones(job.t)]

    tt = cases.tgv(nn, 'senseur', model=model, blocks=blocks, **opts)
    mm = prepare(tt)
    zs = []
    for zz, m in zip(Internal.getZones(tt), mm):
        p, pr, _ = zone_arrays(zz)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(zz), m[0], wall_distance(zz) if mode == 'sa' else None))
        if threads > 1:
            zs[-1].lu_blocks = FastC.lu_thread_blocks(p, threads)
            assert len(zs[-1].lu_blocks) == threads
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in connect.periodic_block_joins(blocks, (n, n, n))])
    s.fill_ghostcells()
    for it in range(nsteps):
        s.step()
    ok = True
    for g, b in zip(got, job.blocks_here):
        if mode == 'sa':    # the device pow of the destruction term differs from the host pow in the last bit
            okb = bool(rel_err(g, zs[b].ro, int(zs[b].pint[41]), neq=6).max() < 1e-12)
        else:
            okb = np.array_equal(g, zs[b].ro)
        if not okb:
            print("rank %d block %d: max abs diff %g" % (rank, b, np.abs(g - zs[b].ro).max()), flush=True)
        ok = ok and okb
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MPI_GPU_WORKER_RESULT %d" % int(flag.item()), flush=True)
    FastC.HOOK['ctx'].close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
