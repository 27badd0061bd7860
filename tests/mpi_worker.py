"""One rank of the multi-rank halo-exchange test (started by test_mpi_gloo.py; gloo, CPU).

The exchange schedule under test is the product's (fast_amd.Mpi.HaloPlan / HaloExchange over
torch.distributed); the zone update between the messages is done by the CPU restatement, which
stands in for the GPU here.  Every rank also advances the whole multi-block job serially and
compares its own block bit for bit.

    python mpi_worker.py <bi> <bj> <bk> <niv> <njv> <nkv> <nsteps>
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
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
import fast_amd.Mpi as FastSmpi
from oracle import oracle as O
from common import prepare, zone_arrays

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def main():
    blocks = tuple(int(a) for a in sys.argv[1:4])
    nv = tuple(int(a) for a in sys.argv[4:7])
    nsteps = int(sys.argv[7])
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    nb = blocks[0] * blocks[1] * blocks[2]
    assert nb == world
    nn = tuple(nv[d] * blocks[d] for d in range(3))
    names = ['cart%d' % b for b in range(nb)]

    # ---- this rank's block, its donor-side joins, the graph ----
    t = cases.tgv(nn, 'senseur', blocks=blocks, only=[rank], extra_numz=SINGLE)
    tc = C.node2Center(t)
    all_joins = connect.periodic_block_joins(blocks, nv)
    connect.add_joins_to_tc(tc, [j for j in all_joins if j[1] == rank], names)
    graph = connect.graph_of(blocks, names, list(range(nb)))
    metrics = prepare(t)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z), metrics[0][0])
    plan = FastSmpi.HaloPlan(FastS.joins_from_tc(t, tc), graph['procDict'], rank, world)
    assert not plan.local
    halo = FastSmpi.HaloExchange(plan, torch.device('cpu'))
    nd = oz.ndimdx

    def pack_from(arr):
        def pack(peer, buf):
            out = buf.numpy()
            off = 0
            for (dn, dpts) in plan.sends[peer]:
                assert dn == names[rank]
                for v in range(5):
                    out[off + v * dpts.size: off + (v + 1) * dpts.size] = arr[v * nd + dpts]
                off += 5 * dpts.size
        return pack

    def unpack_to(arr):
        def unpack(peer, buf):
            src = buf.numpy()
            off = 0
            for (rn, rpts) in plan.recvs[peer]:
                assert rn == names[rank]
                for v in range(5):
                    arr[v * nd + rpts] = src[off + v * rpts.size: off + (v + 1) * rpts.size]
                off += 5 * rpts.size
        return unpack

    halo.run(pack_from(oz.ro), unpack_to(oz.ro))
    oz.fill_ghostcells()
    for it in range(nsteps):
        for nstep in (1, 2, 3):
            oz.stage(nstep)
            out = oz.roles(nstep)[1]
            halo.run(pack_from(out), unpack_to(out))
            oz.bcs(nstep)
        oz.rotate()

    # ---- serial run of the whole job ----
    tt = cases.tgv(nn, 'senseur', blocks=blocks, extra_numz=SINGLE)
    mm = prepare(tt)
    zs = []
    for zz, m in zip(Internal.getZones(tt), mm):
        p, pr, _ = zone_arrays(zz)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(zz), m[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in all_joins])
    s.fill_ghostcells()
    for it in range(nsteps):
        s.step()
    ok = np.array_equal(oz.ro, zs[rank].ro)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MPI_WORKER_RESULT %d peers=%s" % (int(flag.item()), plan.peers()), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
