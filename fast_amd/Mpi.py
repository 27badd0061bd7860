"""MI355X counterpart of ``FastS.Mpi``: one process per GPU, blocks partitioned over the
ranks, ghost cells of match joins that cross ranks exchanged with point-to-point messages.

    (t, tc, metrics) = warmup(t, tc, graph)            Fast/FastS/FastS/Mpi.py:713-945
    _compute(t, metrics, nitrun, tc, graph)            Fast/FastS/FastS/Mpi.py:33-251

``t`` / ``tc`` hold the zones of this rank only; ``graph['procDict']`` maps every zone name to
its rank (Fast/FastS/test/tgvMpi.py:198-256).  The transport is RCCL send/recv over xGMI inside
libfasts_hip.so (fs_comm_*): per sub-iteration and per peer, one packed buffer of donor values
goes out and one comes in (the reference packs per destination rank as well,
Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:944-1582, but ships the receiver indices with
every message; here they are exchanged once, at warmup).  The whole sub-iteration loop, messages
included, then runs inside one native call (fs_compute).

Rank, job size and device come from the environment every launcher of one-process-per-GPU jobs
sets: RANK, WORLD_SIZE, LOCAL_RANK.  The 128-byte communicator identifier made by rank 0 reaches
the other ranks through a file (see ``comm_id_path``); pass ``comm_id=`` to ``warmup`` to carry
it another way (an MPI broadcast, for jobs that span nodes without a shared /tmp).

A caller-owned transport can replace RCCL (``transport=`` of ``warmup``): the sub-iteration is
then driven from here through the split native calls (fs_stage_part, fs_halo_pack,
fs_halo_unpack) and the object moves the packed buffers.  Which joins stay on the rank and which
become messages (``HaloPlan``) does not depend on the transport.
"""
import os
import time
import numpy

from . import FastC
from . import _capi
from . import PyTree as FastS
from .FastC import Internal

__all__ = ['warmup', '_compute', 'HaloPlan', 'register_plan', 'tgv_job', 'rank', 'size', 'local_rank', 'global_ids',
           'comm_id_path', 'barrier', 'allreduce']


def rank():
    return int(os.environ.get('RANK', '0'))


def size():
    return int(os.environ.get('WORLD_SIZE', '1'))


def local_rank():
    return int(os.environ.get('LOCAL_RANK', '0'))


def global_ids(procDict):
    """Job-wide zone numbers: position of the name in the sorted list of every zone of the job."""
    return {name: n for n, name in enumerate(sorted(procDict))}


class HaloPlan:
    """Who sends what to whom.  Built from the donor-side join list of this rank:

        local  : [(rcv_zone, donor_zone, rcv_pts, donor_pts)]            both zones on this rank
        sends  : {peer: [(donor_zone, donor_pts, rcv_zone, rcv_pts)]}    packed in this order
        recvs  : {peer: [(rcv_zone, rcv_pts)]}                           unpacked in this order

    The receive side is learnt from the peers: by the native library when RCCL is the transport
    (fs_comm_setup; `recvs` stays empty here), or through `all_gather_object(out_list, obj)` for a
    caller-owned transport.  loopback: joins between two zones of this rank also travel as
    messages, from the rank to itself (exercises the transport on a single GPU)."""

    def __init__(self, joins, procDict, my_rank, world, all_gather_object=None, loopback=False):
        self.rank = my_rank
        self.local = []
        self.sends = {}
        self.recvs = {}
        outgoing = []
        for (rn, dn, rpts, dpts) in joins:
            owner = procDict.get(rn, my_rank) if procDict else my_rank
            if procDict and procDict.get(dn, my_rank) != my_rank:
                raise ValueError("HaloPlan: donor zone %s of a join is not on rank %d" % (dn, my_rank))
            if owner == my_rank and not loopback:
                self.local.append((rn, dn, rpts, dpts))
            else:
                if owner < 0 or owner >= world:
                    raise ValueError("HaloPlan: zone %s lives on rank %d but the job has %d ranks" % (rn, owner, world))
                rp = numpy.ascontiguousarray(rpts, dtype=numpy.int32)
                self.sends.setdefault(owner, []).append((dn, numpy.ascontiguousarray(dpts, dtype=numpy.int32), rn, rp))
                outgoing.append((owner, rn, rp))
        if all_gather_object is not None:
            gathered = [None] * world
            all_gather_object(gathered, outgoing)
            for src in range(world):
                for (dest, rn, rpts) in gathered[src]:
                    if dest == my_rank:
                        self.recvs.setdefault(src, []).append((rn, rpts))

    def peers(self):
        return sorted(set(self.sends) | set(self.recvs))

    def count(self, peer, send, neq=5):
        lst = self.sends.get(peer, []) if send else self.recvs.get(peer, [])
        return sum(int(e[1].size) for e in lst) * neq


def comm_id_path():
    """File through which rank 0 hands the communicator identifier to the other ranks of the node:
    FASTS_HIP_ID_FILE, or a name made only of what every rank of one launch shares whatever process started it
    (MASTER_ADDR, MASTER_PORT and the launcher's run id if it sets one), in a directory of the user's own."""
    p = os.environ.get('FASTS_HIP_ID_FILE')
    if p:
        return p
    job = '%s_%s_%s' % (os.environ.get('MASTER_ADDR', 'local'), os.environ.get('MASTER_PORT', '0'),
                        os.environ.get('TORCHELASTIC_RUN_ID', os.environ.get('SLURM_JOB_ID', '0')))
    job = ''.join(ch if ch.isalnum() or ch in '._-' else '_' for ch in job)
    d = os.path.join(os.environ.get('XDG_RUNTIME_DIR') or os.environ.get('TMPDIR') or '/tmp', 'fasts_hip_%d' % os.getuid())
    return os.path.join(d, 'comm_id_' + job)


def _share_comm_id(r, timeout=300.):
    """Rank 0 writes the identifier into a file it creates itself (O_EXCL, mode 0600, in a directory only the user can
    write: a file left by another job or planted by someone else is never read as this job's); the other ranks wait
    for a file that is complete and belongs to the user."""
    path = comm_id_path()
    d = os.path.dirname(path) or '.'
    if r == 0:
        os.makedirs(d, mode=0o700, exist_ok=True)
        cid = _capi.comm_unique_id()
        try:
            os.remove(path)            # the leftover of an earlier job of this name
        except OSError:
            pass
        tmp = '%s.%d' % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(cid)
        os.replace(tmp, path)      # appears whole or not at all
        return cid
    t0 = time.time()
    while True:
        try:
            st = os.stat(path)
            # the file must be the user's own, and not older than the launcher of this rank: an older one is the leftover
            # of an earlier job that used the same name (rank 0 removes it before it writes, and after the ranks have met)
            if st.st_uid == os.getuid() and st.st_mtime >= _process_start() - 1.:
                with open(path, 'rb') as f:
                    cid = f.read()
                if len(cid) == _capi.COMM_ID_BYTES:
                    return cid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise _capi.FastsHipError("fast_amd.Mpi: rank %d waited %.0f s for the communicator id in %s "
                                      "(set FASTS_HIP_ID_FILE to a path every rank can reach)" % (r, timeout, path))
        time.sleep(0.02)


_T_IMPORT = time.time()


def _process_start():
    """Start of the launcher of this rank (its parent), which comes before the ranks of the job exist; this process's
    own start where that cannot be read."""
    best = _T_IMPORT
    for pid in (os.getpid(), os.getppid()):
        try:
            best = min(best, os.stat('/proc/%d' % pid).st_mtime)
        except OSError:
            pass
    return best


class _Rccl:
    """Transport inside the native library."""
    native = True

    def __init__(self, r, n, comm_id=None):
        self.rank, self.world, self.comm_id = r, n, comm_id

    def init(self, ctx):
        cid = self.comm_id if self.comm_id is not None else _share_comm_id(self.rank)
        ctx.comm_init(self.rank, self.world, cid)      # collective: every rank has the id once it returns
        if self.comm_id is None and self.rank == 0:
            try:
                os.remove(comm_id_path())
            except OSError:
                pass

    def register(self, ctx, plan, names, gids):
        for p in plan.peers():
            for (dn, dpts, rn, rpts) in plan.sends.get(p, []):
                ctx.send_add_remote(names[dn], p, dpts, gids[rn], rpts)

    def ready(self, ctx):
        ctx.comm_setup()

    def exchange(self, ctx, nstep):
        ctx.halo_exchange(nstep)


def register_plan(ctx, plan, names):
    """Join tables of a plan for a caller-owned transport: local joins, the donor lists packed for each peer
    and the ghost lists filled from each peer (peer tag = peer rank)."""
    for (rn, dn, rpts, dpts) in plan.local:
        ctx.join_add(names[rn], names[dn], rpts, dpts)
    for p in plan.peers():
        for (dn, dpts, _, _) in plan.sends.get(p, []):
            ctx.send_add(names[dn], p, dpts)
        for (rn, rpts) in plan.recvs.get(p, []):
            if rn not in names:
                raise ValueError("rank %d sends ghost values for zone %s which is not on rank %d" % (p, rn, plan.rank))
            ctx.join_add(names[rn], -p - 1, rpts, None)


def warmup(t, tc, graph=None, infos_ale=None, Adjoint=False, tmy=None, list_graph=None, Padding=None,
           verbose=0, device=None, comm_id=None, transport=None, loopback=False):
    """Same as FastS.PyTree.warmup for the zones of this rank, plus the halo plan and the communicator.

    transport: None = RCCL inside the library.  Otherwise an object with `rank`, `world`,
    `all_gather_object(out, obj)`, `attach(ctx)` (streams, before the upload), `ready(ctx, plan)` (after the upload: message buffers), `mark(ctx)`
    (the donor cells of the sub-iteration are final on the context's stream) and `exchange(ctx, plan, nstep)`
    (pack -> messages -> unpack, complete on return as far as the context's stream is concerned)."""
    if infos_ale is not None or Adjoint or tmy is not None:
        raise ValueError("fast_amd.Mpi.warmup: ALE, adjoint and running statistics are not supported")
    if device is None:
        device = local_rank()
    tr = transport if transport is not None else _Rccl(rank(), size(), comm_id)
    r, n = tr.rank, tr.world
    procDict = graph['procDict'] if graph is not None else None
    zones, metrics, dtloc = FastS._warmup_host(t, tc, Padding, device)
    ctx, names = FastS._new_context(zones, metrics, dtloc, device)
    ctx.set_first_it(FastC.FIRST_IT)
    native = getattr(tr, 'native', False)
    flat = FastS.flat_tc(tc) if native else None
    plan = HaloPlan([] if flat is not None else FastS.joins_from_tc(t, tc), procDict, r, n,
                    None if native else tr.all_gather_object, loopback)
    messages = native and (n > 1 or bool(plan.sends) or (flat is not None and loopback))
    if flat is not None:
        # flattened connectivity: blocks for this rank become local joins, the others sends; the receiver zone
        # numbers in the arrays are indices on the destination rank, so a zone's global id is its own index
        for nd in names.values():
            ctx.zone_set_global_id(nd, nd)
        if messages:
            tr.init(ctx)
        ctx.joins_set(flat[0], flat[1], -1 if loopback else r)   # loopback: no block counts as local
    elif native:
        for (rn, dn, rpts, dpts) in plan.local:
            ctx.join_add(names[rn], names[dn], rpts, dpts)
        if messages:
            gids = global_ids(procDict if procDict else names)
            for zn, nd in names.items():
                ctx.zone_set_global_id(nd, gids[zn])
            tr.init(ctx)
            tr.register(ctx, plan, names, gids)
    else:
        tr.attach(ctx)
        register_plan(ctx, plan, names)
    ctx.upload()
    hook = FastS._set_hook(ctx, dtloc, names, mpi=1)
    hook['plan'] = plan
    hook['transport'] = tr
    hook['messages'] = messages
    # ghost cells of the initial field (Fast/FastS/FastS/Mpi.py:934)
    if native:
        if messages:
            tr.ready(ctx)
            tr.exchange(ctx, -_capi.FS_LEVEL_N)
    else:
        tr.ready(ctx, plan)
        tr.mark(ctx)
        tr.exchange(ctx, plan, -_capi.FS_LEVEL_N)
    ctx.fill_ghostcells(_capi.FS_LEVEL_N)
    ctx.download(_capi.FS_LEVEL_N)
    return (t, tc, metrics)


def _is_implicit(zones):
    return int(Internal.getNodeFromName2(zones[0], 'Parameter_int')[1][29]) < 2


def _steps_split(ctx, tr, plan, nitrun, nitmax, nit=1, implicit=False):
    """Sub-iteration loop over a caller-owned transport, the schedule fs_compute runs natively: the tiles that
    feed messages first, the messages while the rest of the stage runs, then local joins and BCs.  implicit:
    the last of the nitmax slots is the reference's verification pass (Fast/FastS/FastS/PyTree.py:123)."""
    for it in range(nit):
        for nstep in range(1, nitmax + 1):
            n = nitrun * nit + it
            if implicit and nstep == nitmax:
                ctx.stage(n, nstep)
                continue
            ctx.stage_part(n, nstep, 0)
            tr.mark(ctx)
            ctx.stage_part(n, nstep, 1)
            tr.exchange(ctx, plan, nstep)
            ctx.local_joins(nstep)
            ctx.bcs(nstep)
        if implicit or nitmax % 2 == 1:
            ctx.rotate()


def _advance(nitrun, nitmax, nit, implicit):
    ctx = FastS._ctx()
    tr = FastC.HOOK.get('transport')
    if tr is None:
        raise _capi.FastsHipError("fast_amd.Mpi._compute: call fast_amd.Mpi.warmup first")
    if getattr(tr, 'native', False):
        ctx.compute(nitrun, 1, nitmax, nit)
    else:
        _steps_split(ctx, tr, FastC.HOOK['plan'], nitrun, nitmax, nit, implicit)


def _compute(t, metrics, nitrun, tc=None, graph=None, tc2=None, graph2=None, layer="c", NIT=1,
             ucData=None, vtune=None, graphInvIBCD_WM=None):
    """Compute NIT time steps; sub-iteration = RHS + update -> halo messages -> local joins -> BCs
    (Fast/FastS/FastS/Mpi.py:117-233)."""
    if tc2 is not None or ucData is not None:
        raise ValueError("fast_amd.Mpi._compute: tc2 / ucData are not supported")
    ctx = FastS._ctx()
    dtloc = FastC.HOOK['dtloc']
    nitmax = int(dtloc[0])
    nit = 1 if layer == "Python" else NIT
    implicit = _is_implicit(Internal.getZones(t))
    _advance(nitrun, nitmax, nit, implicit)
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


def allreduce(vals, op='sum'):
    """vals reduced over the ranks of the job, on every rank (one rank: a copy)."""
    if FastC.HOOK is not None and FastC.HOOK.get('messages') and FastC.HOOK['transport'].world > 1:
        return FastS._ctx().allreduce(vals, op)
    if size() > 1:
        raise _capi.FastsHipError("fast_amd.Mpi.allreduce: call fast_amd.Mpi.warmup first")
    return numpy.atleast_1d(numpy.asarray(vals, dtype=numpy.float64)).copy()


def barrier():
    """Every rank's device work so far is complete when this returns."""
    FastS._ctx().sync()
    allreduce([0.])


def _computeEnstrophy(t, metrics):
    """(enstrophy, kinetic energy) per unit volume of the whole job (Fast/FastS/FastS/PyTree.py:1869-1882, summed
    over the ranks like the reference's MPI reductions, Fast/FastS/FastS/Mpi.py:954)."""
    s = allreduce(FastS._ctx().tgv_sums(_capi.FS_LEVEL_N), 'sum')
    return (s[0] / s[2], s[1] / s[2])


class tgv_job:
    """Weak-scaling Taylor-Green job: the periodic box is split in `blocks` equal blocks of n^3 cells (n: one number,
    or the three cell counts of a block), dealt to the ranks in consecutive runs of blocks/world (one block per rank
    is the layout of Fast/FastS/test/tgvMpi.py:29-50; several per rank gives local joins next to the messages)."""

    def __init__(self, n, blocks, scheme, model, my_rank=None, world=None, local_rank_=None, time_step=None,
                 transport=None, loopback=False, flat=False, **case_options):
        from . import cases, connect
        from . import pytree_lite as C
        my_rank = rank() if my_rank is None else my_rank
        world = size() if world is None else world
        nb = blocks[0] * blocks[1] * blocks[2]
        if nb % world:
            raise ValueError("tgv_job: %d blocks cannot be dealt evenly to %d ranks" % (nb, world))
        per = nb // world
        owner = [b // per for b in range(nb)]
        mine = [b for b in range(nb) if owner[b] == my_rank]
        nloc = (n, n, n) if isinstance(n, int) else tuple(int(v) for v in n)
        nn = (nloc[0] * blocks[0], nloc[1] * blocks[1], nloc[2] * blocks[2])
        if time_step is None:
            time_step = 0.003 * 32. / max(nn)
        self.t = cases.tgv(nn, scheme, model=model, time_step=time_step, blocks=blocks, only=mine, **case_options)
        names = ['cart%d' % b for b in range(nb)]
        self.tc = C.node2Center(self.t)
        joins = [j for j in connect.periodic_block_joins(blocks, nloc) if owner[j[1]] == my_rank]
        connect.add_joins_to_tc(self.tc, joins, names)
        self.graph = connect.graph_of(blocks, names, owner)
        if flat:     # hand the joins over as the flattened connectivity arrays (tc/Parameter_int, tc/Parameter_real)
            local = {names[b]: mine.index(b) for b in mine}
            on_owner = {names[b]: b % per for b in range(nb)}
            pint, preal = connect.flatten_joins(FastS.joins_from_tc(self.t, self.tc), local, self.graph['procDict'],
                                                my_rank, on_owner)
            connect.add_flat_to_tc(self.tc, pint, preal)
        self.blocks_here = mine
        (self.t, self.tc, self.metrics) = warmup(self.t, self.tc, self.graph, device=local_rank_,
                                                 transport=transport, loopback=loopback)
        self.ctx = FastC.HOOK['ctx']
        self.zones = Internal.getZones(self.t)
        self.zone = self.zones[0]
        self.nitmax = int(FastC.HOOK['dtloc'][0])
        self.implicit = _is_implicit([self.zone])

    def step(self, it):
        _advance(it, self.nitmax, 1, self.implicit)
