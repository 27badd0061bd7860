"""Several zones on the device: local match joins in one context, and the split calls
(fs_stage / fs_halo_pack / fs_halo_unpack / fs_bcs) between two contexts that stand for two
ranks.  Checked bit for bit against the multi-zone CPU restatement."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal, _capi
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
import fast_amd.Mpi as FastSmpi
from oracle import oracle as O
from common import zone_arrays

pytestmark = pytest.mark.gpu

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def _oracle_job(nn, blocks, scheme, model):
    from common import prepare
    t = cases.tgv(nn, scheme, model=model, blocks=blocks, extra_numz=SINGLE)
    m = prepare(t)
    zs = []
    for z, mm in zip(Internal.getZones(t), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in connect.periodic_block_joins(blocks, nv)])
    s.fill_ghostcells()
    return s


@pytest.mark.parametrize("blocks,nn,scheme,model", [
    ((1, 1, 2), (16, 9, 12), 'senseur', 'NSLaminar'),
    ((2, 2, 1), (12, 12, 7), 'ausmpred', 'NSLaminar'),
    ((2, 2, 2), (132, 8, 8), 'senseur', 'Euler'),          # blocks wider than one 64-cell tile
])
@pytest.mark.parametrize("layer", ["c", "Python", "flat"])
def test_local_joins_bit_exact(blocks, nn, scheme, model, layer):
    s = _oracle_job(nn, blocks, scheme, model)
    t = cases.tgv(nn, scheme, model=model, blocks=blocks, extra_numz=SINGLE)
    zones = Internal.getZones(t)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, nv), [z[0] for z in zones])
    if layer == "flat":      # the same joins through the flattened connectivity arrays and fs_joins_set
        index = {z[0]: n for n, z in enumerate(zones)}
        connect.add_flat_to_tc(tc, *connect.flatten_joins(FastS.joins_from_tc(t, tc), index, None))
        layer = "c"
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for z, oz in zip(zones, s.zones):
        assert np.array_equal(FastC.level_buffer(z), oz.ro)
    for it in range(1, 4):
        FastS._compute(t, metrics, it, tc, layer=layer)
        s.step()
        for z, oz in zip(zones, s.zones):
            assert np.array_equal(FastC.level_buffer(z), oz.ro), "step %d zone %s" % (it, z[0])


def _two_plans(blocks, nv, names):
    """HaloPlans of ranks 0 and 1 without a process group: the all-gather is emulated."""
    all_joins = connect.periodic_block_joins(blocks, nv)
    procDict = {names[b]: b for b in range(2)}
    store = {}

    def gather_for(rank):
        def g(lst, obj):
            store[rank] = obj
            for i in range(len(lst)):
                lst[i] = store.get(i, [])
        return g

    def joins_of(rank):
        return [(names[br], names[bd], r, d) for br, bd, r, d in all_joins if bd == rank]

    for r in range(2):           # first pass fills the store
        FastSmpi.HaloPlan(joins_of(r), procDict, r, 2, all_gather_object=gather_for(r))
    return [FastSmpi.HaloPlan(joins_of(r), procDict, r, 2, all_gather_object=gather_for(r)) for r in range(2)]


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("blocks,nn", [((1, 1, 2), (10, 8, 140)), ((2, 1, 1), (140, 6, 5)), ((1, 2, 1), (40, 36, 9))])
def test_two_contexts_exchange_through_the_split_calls(blocks, nn, split):
    import torch
    s = _oracle_job(nn, blocks, 'senseur', 'NSLaminar')
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    names = ['cart0', 'cart1']
    plans = _two_plans(blocks, nv, names)
    ctxs, trees = [], []
    if FastC.HOOK is not None and FastC.HOOK.get('ctx') is not None:
        FastC.HOOK['ctx'].close(); FastC.HOOK = None
    for r in range(2):
        t = cases.tgv(nn, 'senseur', blocks=blocks, only=[r], extra_numz=SINGLE)
        zones, metrics, dtloc = FastS._warmup_host(t, None)
        ctx = _capi.Context(0)
        pint, preal, _ = zone_arrays(zones[0])
        ctx.zone_add(pint, preal, FastC.level_buffer(zones[0], ''), FastC.level_buffer(zones[0], '_M1'),
                     FastC.level_buffer(zones[0], '_P1'), metrics[0][0])
        FastSmpi.register_plan(ctx, plans[r], {names[r]: 0})
        ctx.upload()
        ctxs.append(ctx); trees.append((t, zones[0]))
    n01 = ctxs[0].halo_count(1, True)
    assert n01 == ctxs[1].halo_count(0, False) == plans[0].count(1, True) > 0
    buf = [torch.empty(n01, dtype=torch.float64, device='cuda:0') for _ in range(2)]
    torch.cuda.synchronize()

    def exchange(nstep):
        ctxs[0].halo_pack(nstep, 1, buf[0].data_ptr(), n01)
        ctxs[1].halo_pack(nstep, 0, buf[1].data_ptr(), n01)
        ctxs[0].sync(); ctxs[1].sync()
        ctxs[1].halo_unpack(nstep, 0, buf[0].data_ptr(), n01)
        ctxs[0].halo_unpack(nstep, 1, buf[1].data_ptr(), n01)

    exchange(-_capi.FS_LEVEL_N)
    for r in range(2):
        ctxs[r].fill_ghostcells(_capi.FS_LEVEL_N)
        ctxs[r].download(_capi.FS_LEVEL_N)
        assert np.array_equal(FastC.level_buffer(trees[r][1]), s.zones[r].ro)
    for it in range(1, 4):
        for nstep in (1, 2, 3):
            for c in ctxs:
                if split:      # donor tiles first, the rest afterwards: same result
                    c.stage_part(it, nstep, 0)
                    c.stage_part(it, nstep, 1)
                else:
                    c.stage(it, nstep)
            exchange(nstep)
            for c in ctxs:
                c.local_joins(nstep); c.bcs(nstep)
            for c in ctxs:
                c.sync()
        s.step()
        for r in range(2):
            ctxs[r].rotate()
            FastC.switchPointers__([trees[r][1]], 1, 3)
            ctxs[r].download(_capi.FS_LEVEL_N)
            assert np.array_equal(FastC.level_buffer(trees[r][1]), s.zones[r].ro), "step %d rank %d" % (it, r)
    with pytest.raises(_capi.FastsHipError):
        ctxs[0].compute(4, 1, 3, 1)      # joins to another context need the split calls
    for c in ctxs:
        c.close()


def test_flat_connectivity_is_validated():
    """fs_joins_set refuses what a match join cannot be, and arrays that end early, with a message."""
    nn, blocks = (8, 8, 16), (1, 1, 2)
    t = cases.tgv(nn, 'senseur', blocks=blocks, extra_numz=SINGLE)
    zones = Internal.getZones(t)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, (8, 8, 8)), [z[0] for z in zones])
    index = {z[0]: n for n, z in enumerate(zones)}
    pint, preal = connect.flatten_joins(FastS.joins_from_tc(t, tc), index, None)
    zs, metrics, dtloc = FastS._warmup_host(t, None)

    def fresh():
        ctx, _ = FastS._new_context(zs, metrics, dtloc, 0)
        return ctx

    ctx = fresh(); ctx.joins_set(pint, preal, 0); ctx.upload(); ctx.close()
    bad = preal.copy(); bad[3] = 0.5                      # an interpolation weight: not a match join
    with pytest.raises(_capi.FastsHipError, match="unit coefficients"):
        fresh().joins_set(pint, bad, 0)
    with pytest.raises(_capi.FastsHipError, match="outside|truncated"):
        fresh().joins_set(pint[:pint.size // 2], preal, 0)
    ech = int(pint[4]); nrac = int(pint[ech + 1])
    ibc = pint.copy(); ibc[ech + 4 + 3 * nrac] = 2        # ibcType >= 0: immersed boundary rac
    with pytest.raises(_capi.FastsHipError, match="immersed"):
        fresh().joins_set(ibc, preal, 0)
    far = pint.copy(); far[ech + 4 + 5 * nrac] = 7        # donor zone that does not exist
    with pytest.raises(_capi.FastsHipError, match="donor zone"):
        fresh().joins_set(far, preal, 0)
