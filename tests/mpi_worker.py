"""One rank of the multi-rank halo-exchange test (started by test_mpi_gloo.py; gloo, CPU).

The join split under test is the product's (fast_amd.Mpi.HaloPlan); the messages travel over gloo
(halo_gloo.HaloExchange); the zone update between the messages is done by the CPU restatement, which
stands in for the GPU here.  Every rank also advances the whole multi-block job serially and
compares its own blocks bit for bit (several blocks per rank: local joins next to messages).

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
from halo_gloo import HaloExchange

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def main():
    blocks = tuple(int(a) for a in sys.argv[1:4])
    nv = tuple(int(a) for a in sys.argv[4:7])
    nsteps = int(sys.argv[7])
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    nb = blocks[0] * blocks[1] * blocks[2]
    assert nb % world == 0
    per = nb // world
    owner = [b // per for b in range(nb)]
    mine = [b for b in range(nb) if owner[b] == rank]
    nn = tuple(nv[d] * blocks[d] for d in range(3))
    names = ['cart%d' % b for b in range(nb)]

    # ---- this rank's blocks, their donor-side joins, the graph ----
    t = cases.tgv(nn, 'senseur', blocks=blocks, only=mine, extra_numz=SINGLE)
    tc = C.node2Center(t)
    all_joins = connect.periodic_block_joins(blocks, nv)
    connect.add_joins_to_tc(tc, [j for j in all_joins if owner[j[1]] == rank], names)
    graph = connect.graph_of(blocks, names, owner)
    metrics = prepare(t)
    ozs = {}
    for z, m in zip(Internal.getZones(t), metrics):
        pint, preal, _ = zone_arrays(z)
        ozs[z[0]] = O.OracleZone(pint, preal, FastC.level_buffer(z), m[0])
    plan = FastSmpi.HaloPlan(FastS.joins_from_tc(t, tc), graph['procDict'], rank, world, dist.all_gather_object)
    assert bool(plan.local) == (per > 1)
    halo = HaloExchange(plan, torch.device('cpu'))

    def pack_from(arr_of):
        def pack(peer, buf):
            out = buf.numpy()
            off = 0
            for (dn, dpts, _, _) in plan.sends[peer]:
                arr, nd = arr_of(dn), ozs[dn].ndimdx
                for v in range(5):
                    out[off + v * dpts.size: off + (v + 1) * dpts.size] = arr[v * nd + dpts]
                off += 5 * dpts.size
        return pack

    def unpack_to(arr_of):
        def unpack(peer, buf):
            src = buf.numpy()
            off = 0
            for (rn, rpts) in plan.recvs[peer]:
                arr, nd = arr_of(rn), ozs[rn].ndimdx
                for v in range(5):
                    arr[v * nd + rpts] = src[off + v * rpts.size: off + (v + 1) * rpts.size]
                off += 5 * rpts.size
        return unpack

    def local_joins(arr_of):
        for (rn, dn, rpts, dpts) in plan.local:
            ar, ad = arr_of(rn), arr_of(dn)
            for v in range(5):
                ar[v * ozs[rn].ndimdx + rpts] = ad[v * ozs[dn].ndimdx + dpts]

    level_n = lambda name: ozs[name].ro
    halo.run(pack_from(level_n), unpack_to(level_n))
    local_joins(level_n)
    for oz in ozs.values():
        oz.fill_ghostcells()
    for it in range(nsteps):
        for nstep in (1, 2, 3):
            for oz in ozs.values():
                oz.stage(nstep)
            out = lambda name: ozs[name].roles(nstep)[1]
            halo.run(pack_from(out), unpack_to(out))
            local_joins(out)
            for oz in ozs.values():
                oz.bcs(nstep)
        for oz in ozs.values():
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
    ok = all(np.array_equal(ozs[names[b]].ro, zs[b].ro) for b in mine)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MPI_WORKER_RESULT %d peers=%s" % (int(flag.item()), plan.peers()), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
