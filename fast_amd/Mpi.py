"""MI355X counterpart of ``FastS.Mpi``: one process per GPU, blocks partitioned over the
ranks, ghost cells of match joins that cross ranks exchanged with point-to-point messages.

    (t, tc, metrics) = warmup(t, tc, graph)            Fast/FastS/FastS/Mpi.py:713-945
    _compute(t, metrics, nitrun, tc, graph)            Fast/FastS/FastS/Mpi.py:33-251

``t`` / ``tc`` hold the zones of this rank only; ``graph['procDict']`` maps every zone name to
its rank (Fast/FastS/test/tgvMpi.py:198-256).  The transport is ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU): per sub-iteration and per peer,
one packed buffer of donor values goes out and one comes in (the reference packs per
destination rank as well, Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:944-1582, but ships
the receiver indices with every message; here they are exchanged once, at warmup).

The exchange schedule (``HaloPlan`` / ``HaloExchange``) does not depend on the device so that it
can be driven by host arrays in tests; the time step itself only exists on the GPU.
"""
import os
import numpy

from . import FastC
from . import _capi
from . import PyTree as FastS
from .FastC import Internal

__all__ = ['warmup', '_compute', 'HaloPlan', 'HaloExchange', 'tgv_job', 'rank', 'size']


def _dist():
    import torch.distributed as dist
    return dist


def rank():
    d = _dist()
    return d.get_rank() if d.is_available() and d.is_initialized() else 0


def size():
    d = _dist()
    return d.get_world_size() if d.is_available() and d.is_initialized() else 1


class HaloPlan:
    """Who sends what to whom.  Built from the donor-side join list of this rank:

        local  : [(rcv_zone, donor_zone, rcv_pts, donor_pts)]         both zones on this rank
        sends  : {peer: [(donor_zone, donor_pts)]}                    packed in this order
        recvs  : {peer: [(rcv_zone, rcv_pts)]}                        unpacked in this order

    The receive side is learnt from the peers (one all-gather of the receiver index lists)."""

    def __init__(self, joins, procDict, my_rank, world, all_gather_object=None):
        self.rank = my_rank
        self.local = []
        self.sends = {}
        self.recvs = {}
        outgoing = []
        for (rn, dn, rpts, dpts) in joins:
            owner = procDict.get(rn, my_rank) if procDict else my_rank
            if procDict and procDict.get(dn, my_rank) != my_rank:
                raise ValueError("HaloPlan: donor zone %s of a join is not on rank %d" % (dn, my_rank))
            if owner == my_rank:
                self.local.append((rn, dn, rpts, dpts))
            else:
                self.sends.setdefault(owner, []).append((dn, numpy.ascontiguousarray(dpts, dtype=numpy.int32)))
                outgoing.append((owner, rn, numpy.ascontiguousarray(rpts, dtype=numpy.int32)))
        if world > 1:
            if all_gather_object is None:
                all_gather_object = _dist().all_gather_object
            gathered = [None] * world
            all_gather_object(gathered, outgoing)
            for src in range(world):
                if src == my_rank:
                    continue
                for (dest, rn, rpts) in gathered[src]:
                    if dest == my_rank:
                        self.recvs.setdefault(src, []).append((rn, rpts))
        elif outgoing:
            raise ValueError("HaloPlan: joins to other ranks but the job has a single rank")

    def peers(self):
        return sorted(set(self.sends) | set(self.recvs))

    def count(self, peer, send, neq=5):
        lst = self.sends.get(peer, []) if send else self.recvs.get(peer, [])
        return sum(int(p.size) for _, p in lst) * neq


class HaloExchange:
    """Per-peer send / receive buffers and the message round of one sub-iteration."""

    def __init__(self, plan, device, neq=5, sync=None, counts=None):
        """device: where pack / unpack work.  With the "nccl" backend (RCCL) the device buffers go
        on the wire as they are.  With a host transport ("gloo") and device buffers, messages are
        staged through pinned host memory; `sync` (a callable that waits for the pack kernels) is
        then called before the copy out."""
        import torch
        dist = _dist()
        self.plan = plan
        self.sbuf = {}
        self.rbuf = {}
        self.hs = {}
        self.hr = {}
        self.sync = sync
        dev = torch.device(device)
        backend = dist.get_backend() if dist.is_available() and dist.is_initialized() else 'none'
        self.staged = dev.type == 'cuda' and backend != 'nccl'
        # counts(peer, is_send): message sizes as the device context counts them (zones may carry 5 or 6 variables)
        if counts is None:
            counts = lambda p, send: plan.count(p, send, neq)
        for p in plan.peers():
            self.sbuf[p] = torch.empty(counts(p, True), dtype=torch.float64, device=dev)
            self.rbuf[p] = torch.empty(counts(p, False), dtype=torch.float64, device=dev)
            if self.staged:
                self.hs[p] = torch.empty(self.sbuf[p].numel(), dtype=torch.float64).pin_memory()
                self.hr[p] = torch.empty(self.rbuf[p].numel(), dtype=torch.float64).pin_memory()

    def run(self, pack, unpack):
        """pack(peer, send_tensor) fills the outgoing values; all messages of the round are posted
        together; unpack(peer, recv_tensor) scatters what arrived."""
        dist = _dist()
        peers = self.plan.peers()
        if not peers:
            return
        ops = []
        for p in peers:
            if self.sbuf[p].numel():
                pack(p, self.sbuf[p])
        if self.staged:
            if self.sync is not None:
                self.sync()
            for p in peers:
                self.hs[p].copy_(self.sbuf[p])      # stream-ordered after the pack kernels, blocking
        wire_s = self.hs if self.staged else self.sbuf
        wire_r = self.hr if self.staged else self.rbuf
        for p in peers:
            if wire_s[p].numel():
                ops.append(dist.P2POp(dist.isend, wire_s[p], p))
            if wire_r[p].numel():
                ops.append(dist.P2POp(dist.irecv, wire_r[p], p))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if self.staged:
            for p in peers:
                self.rbuf[p].copy_(self.hr[p])
        for p in peers:
            if self.rbuf[p].numel():
                unpack(p, self.rbuf[p])


def register_plan(ctx, plan, names):
    """Hand the join tables of a plan to a device context: local joins, the donor lists packed
    for each peer, and the ghost lists filled from each peer (peer tag = peer rank)."""
    for (rn, dn, rpts, dpts) in plan.local:
        ctx.join_add(names[rn], names[dn], rpts, dpts)
    for p in plan.peers():
        for (dn, dpts) in plan.sends.get(p, []):
            ctx.send_add(names[dn], p, dpts)
        for (rn, rpts) in plan.recvs.get(p, []):
            if rn not in names:
                raise ValueError("rank %d sends ghost values for zone %s which is not on rank %d"
                                 % (p, rn, plan.rank))
            ctx.join_add(names[rn], -p - 1, rpts, None)


def _exchange(ctx, halo, nstep):
    """nstep >= 1: on the output level of that sub-iteration; nstep <= 0: on time level -nstep."""
    import torch
    pack = lambda p, b: ctx.halo_pack(nstep, p, b.data_ptr(), b.numel())
    unpack = lambda p, b: ctx.halo_unpack(nstep, p, b.data_ptr(), b.numel())
    stream = getattr(halo, 'stream', None)
    if stream is None:
        halo.run(pack, unpack)
    else:
        with torch.cuda.stream(stream):
            halo.run(pack, unpack)


def warmup(t, tc, graph=None, infos_ale=None, Adjoint=False, tmy=None, list_graph=None, Padding=None,
           verbose=0, device=None):
    """Same as FastS.PyTree.warmup for the zones of this rank, plus the halo plan."""
    import torch
    if infos_ale is not None or Adjoint or tmy is not None:
        raise ValueError("fast_amd.Mpi.warmup: ALE, adjoint and running statistics are not supported")
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
    r, n = rank(), size()
    procDict = graph['procDict'] if graph is not None else None
    zones, metrics, dtloc = FastS._warmup_host(t, tc, Padding)
    torch.cuda.set_device(device)
    ctx, names = FastS._new_context(zones, metrics, dtloc, device)
    ctx.set_first_it(FastC.FIRST_IT)
    # kernels and RCCL messages are ordered on ONE explicit (non-default) torch stream: the library
    # launches on it (fs_set_stream) and the messages are posted while it is torch's current stream,
    # so RCCL's event hand-shake orders them after the pack kernels and before the unpack kernels
    # A second stream carries pack -> messages -> unpack so that they overlap the tiles of a stage that
    # do not feed a message (fs_stage_part); events order the two streams.
    main = torch.cuda.Stream(device=device)
    comm = torch.cuda.Stream(device=device)
    ctx.set_stream(main.cuda_stream)
    ctx.set_comm_stream(comm.cuda_stream)
    plan = HaloPlan(FastS.joins_from_tc(t, tc), procDict, r, n)
    register_plan(ctx, plan, names)
    ctx.upload()
    hook = FastS._set_hook(ctx, dtloc, names, mpi=1)
    with torch.cuda.stream(comm):
        halo = HaloExchange(plan, torch.device('cuda', device), counts=lambda p, send: ctx.halo_count(p, send))
    halo.stream = comm
    halo.main = main
    halo.overlap = os.environ.get('FASTS_HIP_OVERLAP', '1') != '0'
    hook['halo'] = halo
    # ghost cells of the initial field
    comm.wait_stream(main)
    _exchange(ctx, halo, -_capi.FS_LEVEL_N)
    main.wait_stream(comm)
    ctx.fill_ghostcells(_capi.FS_LEVEL_N)
    ctx.download(_capi.FS_LEVEL_N)
    return (t, tc, metrics)


def _is_implicit(zones):
    return int(Internal.getNodeFromName2(zones[0], 'Parameter_int')[1][29]) < 2


def _steps(ctx, halo, nitrun, nitmax, nit=1, implicit=False):
    """implicit: the last of the nitmax slots is the reference's verification pass and is skipped
    (Fast/FastS/FastS/PyTree.py:123); the Newton iterates travel through the joins after every
    sub-iteration, each zone runs its own LU-SGS sweep (lu_match = 0)."""
    main, comm = getattr(halo, 'main', None), getattr(halo, 'stream', None)
    for it in range(nit):
        for nstep in range(1, nitmax + 1):
            if implicit and nstep == nitmax:
                continue
            n = nitrun * nit + it
            if main is not None and getattr(halo, 'overlap', False):
                ctx.stage_part(n, nstep, 0)          # tiles that feed the messages
                donors_ready = main.record_event()
                ctx.stage_part(n, nstep, 1)          # everything else, overlapped with the messages
                comm.wait_event(donors_ready)
            else:
                ctx.stage(n, nstep)
                if main is not None:
                    comm.wait_stream(main)
            _exchange(ctx, halo, nstep)
            if main is not None:
                main.wait_stream(comm)
            ctx.local_joins(nstep)
            ctx.bcs(nstep)
        if implicit or nitmax % 2 == 1:
            ctx.rotate()


def _compute(t, metrics, nitrun, tc=None, graph=None, tc2=None, graph2=None, layer="c", NIT=1,
             ucData=None, vtune=None, graphInvIBCD_WM=None):
    """Compute NIT time steps; sub-iteration = RHS + update -> halo messages -> local joins -> BCs
    (Fast/FastS/FastS/Mpi.py:117-233)."""
    if tc2 is not None or ucData is not None:
        raise ValueError("fast_amd.Mpi._compute: tc2 / ucData are not supported")
    ctx = FastS._ctx()
    halo = FastC.HOOK.get('halo')
    if halo is None:
        raise _capi.FastsHipError("fast_amd.Mpi._compute: call fast_amd.Mpi.warmup first")
    dtloc = FastC.HOOK['dtloc']
    nitmax = int(dtloc[0])
    nit = 1 if layer == "Python" else NIT
    implicit = _is_implicit(Internal.getZones(t))
    _steps(ctx, halo, nitrun, nitmax, nit, implicit)
    case = nit % 3
    if case != 0:
        FastC.switchPointers__(Internal.getZones(t), case, 3 if implicit else nitmax % 2 + 2)
    FastC.HOOK["FIRST_IT"] = 1
    FastC.FIRST_IT = 1
    if not FastS._lazy():
        ctx.download(_capi.FS_LEVEL_N)
    if vtune is None:
        return None
    return 0., 0., 0.


class tgv_job:
    """Weak-scaling Taylor-Green job: the periodic box is split in `blocks` equal blocks of n^3
    cells, rank r owns block r (the layout of Fast/FastS/test/tgvMpi.py:29-50)."""

    def __init__(self, n, blocks, scheme, model, my_rank, world, local_rank, time_step=None, **case_options):
        from . import cases, connect
        from . import pytree_lite as C
        nb = blocks[0] * blocks[1] * blocks[2]
        if nb != world:
            raise ValueError("tgv_job: %d blocks for %d ranks" % (nb, world))
        nn = (n * blocks[0], n * blocks[1], n * blocks[2])
        if time_step is None:
            time_step = 0.003 * 32. / max(nn)
        self.t = cases.tgv(nn, scheme, model=model, time_step=time_step, blocks=blocks, only=[my_rank], **case_options)
        names = ['cart%d' % b for b in range(nb)]
        self.tc = C.node2Center(self.t)
        joins = [j for j in connect.periodic_block_joins(blocks, (n, n, n)) if j[1] == my_rank]
        connect.add_joins_to_tc(self.tc, joins, names)
        self.graph = connect.graph_of(blocks, names, list(range(nb)))
        (self.t, self.tc, self.metrics) = warmup(self.t, self.tc, self.graph, device=local_rank)
        self.ctx = FastC.HOOK['ctx']
        self.halo = FastC.HOOK['halo']
        self.zone = Internal.getZones(self.t)[0]
        self.nitmax = int(FastC.HOOK['dtloc'][0])
        self.implicit = _is_implicit([self.zone])

    def step(self, it):
        _steps(self.ctx, self.halo, it, self.nitmax, 1, self.implicit)
