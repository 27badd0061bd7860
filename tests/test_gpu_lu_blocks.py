"""GPU parity of the LU-SGS solve cut in the blocks of the reference's OpenMP threads (fs_zone_set_lu_blocks) against
the CPU restatement, itself bit-exact against invlu_ of the compiled reference called once per thread block
(test_lu_thread_blocks.py).  The device advances every block by one hyperplane per launch."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal, _capi
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, wall_distance

pytestmark = pytest.mark.gpu

FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}


@pytest.fixture
def threads():
    """Sets the thread count the host model hands to warmup and restores it."""
    old = FastC.OMP_NUM_THREADS

    def set_(n):
        FastC.OMP_NUM_THREADS = n
    yield set_
    FastC.OMP_NUM_THREADS = old


CASES = [
    # n, scheme, model, perturb, bcs, ss_iteration, threads, expected blocks[, nb_relax]
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 3, 4, 4),
    ((12, 24, 33), 'senseur', 'NSLaminar', 0., None, 3, 6, 6),
    ((40, 33, 47), 'ausmpred', 'NSLaminar', 0., None, 2, 12, 12),      # blocks of unequal sizes, several chunks
    ((12, 23, 21), 'roe', 'Euler', 0.15, None, 3, 4, 4),               # curvilinear
    ((35, 31, 10), 'senseur', 'NSLaminar', 0.15, None, 2, 3, 3),
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 3, 4, 4),
    ((11, 22, 23), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 3, 4, 4),
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., FAR, 3, 2, 2),
    ((12, 9, 10), 'ausmpred', 'NSLaminar', 0., None, 3, 8, 1),         # too small to cut: one block
    ((12, 24, 22), 'ausmpred', 'NSTurbulent', 0., {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 4, 4),
    ((12, 22, 21), 'roe', 'NSTurbulent', 0.15, None, 3, 4, 4),
    ((34, 22), 'ausmpred', 'NSLaminar', 0.15, None, 3, 4, 4),          # two-dimensional: cut along j and i
    ((45, 23), 'senseur', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 6, 6),
    # rows that start on 128-byte lines (padded row length a multiple of 16 cells): units shifted onto the lines
    ((28, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 3, 4, 4),
    ((44, 33, 22), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 2, 6, 6),
    ((28, 24, 22), 'ausmpred', 'NSLaminar', 0., {'imin': 'BCWall', 'imax': 'BCWall'}, 3, 4, 4),
    ((60, 44), 'roe', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 8, 8),
    ((28, 24, 22), 'senseur', 'NSLaminar', 0., None, 3, 4, 4, 2),
    # rows that start 4, 8 or 12 cells into a line, line after line: units shifted by a growing number of cells
    ((24, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 3, 4, 4),
    ((32, 24, 22), 'senseur', 'NSLaminar', 0., {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 3, 4, 4),
    ((16, 22, 22), 'roe', 'NSLaminar', 0.15, None, 3, 4, 4),
    ((64, 24, 33), 'ausmpred', 'NSLaminar', 0., {'imin': 'BCWall', 'imax': 'BCWall'}, 2, 6, 6),
    ((48, 44), 'senseur', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 4, 4),
    ((24, 24, 22), 'senseur', 'NSLaminar', 0., None, 3, 4, 4, 2),
    ((32, 24, 22), 'ausmpred', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 4, 4, 2),
    # long lines, few of them per block: the growing shift adds under 15 % of hyperplanes and is the library's own choice
    ((128, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 2, 4, 4),
    ((128, 24, 22), 'senseur', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWall'}, 2, 4, 4, 2),
    # LU-SSOR relaxations block by block
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 3, 4, 4, 2),
    ((40, 33, 47), 'senseur', 'NSLaminar', 0., None, 2, 12, 12, 2),
    ((12, 23, 21), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 3, 4, 4, 3),
    ((12, 24, 22), 'roe', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 4, 4, 2),
    ((45, 23), 'senseur', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 3, 6, 6, 2),
]
CASES = [c if len(c) == 9 else c + (1,) for c in CASES]


@pytest.mark.parametrize("align", [1, 2])
@pytest.mark.parametrize("n,scheme,model,perturb,bcs,ssit,nthreads,nblocks,nb_relax", CASES)
def test_block_lusgs_matches_oracle(threads, monkeypatch, n, scheme, model, perturb, bcs, ssit, nthreads, nblocks, nb_relax, align):
    """align: 1 = units shifted onto 128-byte lines where the library finds it pays, 2 = wherever the layout allows
    (the general case, with a shift that grows from line to line)."""
    monkeypatch.setenv('FASTS_HIP_LU_ALIGN', str(align))
    make = cases.tgv if len(n) == 3 else cases.tgv2d
    t = make(n, scheme, model=model, extra_numz=dict(SINGLE, nb_relax=nb_relax), extra_numb={'ss_iteration': ssit}, bcs=bcs,
             temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    threads(nthreads)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    blocks = FastC.lu_thread_blocks(pint, nthreads)
    assert len(blocks) == nblocks
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.lu_blocks = blocks
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if neq == 5:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
        else:    # the device power function of the Spalart-Allmaras source term is not the host's
            e = rel_err(got, oz.ro, int(pint[41]), neq=6)
            assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4
    if nblocks > 1:
        # the blocks matter: the single sweep of the zone gives another field
        one = O.OracleZone(pint, preal, ro0.copy(), metrics[0][0], wall_distance(z))
        one.ro_m1[:] = one.ro; one.ro_p1[:] = one.ro
        for it in range(1, 5):
            one.step()
        assert not np.array_equal(one.ro, oz.ro)


@pytest.mark.parametrize("align", [1, 2])
def test_block_lusgs_at_size(threads, monkeypatch, align):
    """96^3 cells in the 2 x 8 blocks of 16 threads (96 x 48 x 12 cells each: several waves per hyperplane and block,
    grids of a few thousand workgroups), two time steps of three Newton sub-iterations, bit for bit."""
    monkeypatch.setenv('FASTS_HIP_LU_ALIGN', str(align))
    n = (96, 96, 96)
    t = cases.tgv(n, 'ausmpred', extra_numz=SINGLE, extra_numb={'ss_iteration': 3}, temporal_scheme='implicit',
                  time_step=0.02)
    threads(16)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    blocks = FastC.lu_thread_blocks(pint, 16)
    assert len(blocks) == 16 and tuple(blocks[0]) == (1, 96, 1, 48, 1, 12)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.lu_blocks = blocks
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    for it in range(1, 3):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))


@pytest.mark.parametrize("n", [(40, 33, 47), (12, 24, 22), (96, 25, 9), (60, 44), (130, 21)])
def test_native_thread_decomposition(n):
    """fs_set_lu_threads cuts the zones as FastC.lu_thread_blocks does (itself equal to indice_boucle_lu_ of the compiled
    reference, test_lu_thread_blocks.py), for every thread count up to 40 and a few larger ones."""
    from common import prepare
    make = cases.tgv if len(n) == 3 else cases.tgv2d
    t = make(n, 'ausmpred', extra_numz=SINGLE, extra_numb={'ss_iteration': 2}, temporal_scheme='implicit', time_step=0.02)
    metrics = prepare(t)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    lv = [FastC.level_buffer(z, s) for s in ('', '_M1', '_P1')]
    ctx = _capi.Context(0)
    ctx.zone_add(pint, preal, lv[0], lv[1], lv[2], metrics[0][0])
    assert len(ctx.zone_lu_blocks(0)) == 0
    for nth in list(range(1, 41)) + [48, 64, 96, 128, 200]:
        ctx.set_lu_threads(nth)
        want = FastC.lu_thread_blocks(pint, nth)
        got = ctx.zone_lu_blocks(0)
        if len(want) == 1:
            assert len(got) == 0, nth          # one block: the single sweep
        else:
            assert np.array_equal(got, want), nth
    ctx.close()


def test_lu_blocks_are_validated():
    t = cases.tgv((12, 24, 22), 'ausmpred', extra_numz=SINGLE, extra_numb={'ss_iteration': 2},
                  temporal_scheme='implicit', time_step=0.02)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    lv = [FastC.level_buffer(z, s) for s in ('', '_M1', '_P1')]
    ctx = _capi.Context(0)
    ctx.zone_add(pint, preal, lv[0], lv[1], lv[2], metrics[0][0])
    good = FastC.lu_thread_blocks(pint, 4)
    ctx.zone_set_lu_blocks(0, good)
    with pytest.raises(_capi.FastsHipError, match="overlap"):
        bad = good.copy(); bad[1, 2] -= 1
        ctx.zone_set_lu_blocks(0, bad)
    with pytest.raises(_capi.FastsHipError, match="cover"):
        ctx.zone_set_lu_blocks(0, good[:3])
    with pytest.raises(_capi.FastsHipError, match="interior"):
        bad = good.copy(); bad[0, 0] = 0
        ctx.zone_set_lu_blocks(0, bad)
    ctx.zone_set_lu_blocks(0, None)          # back to one sweep
    ctx.close()
    # lu_match = 1 with thread blocks is not built
    t = cases.tgv((12, 24, 22), 'ausmpred', extra_numz=dict(SINGLE, lu_match=1), extra_numb={'ss_iteration': 2},
                  temporal_scheme='implicit', time_step=0.02)
    old = FastC.OMP_NUM_THREADS
    FastC.OMP_NUM_THREADS = 4
    try:
        with pytest.raises(_capi.FastsHipError, match="lu_match"):
            FastS.warmup(t, None)
    finally:
        FastC.OMP_NUM_THREADS = old
