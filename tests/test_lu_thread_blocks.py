"""LU-SGS solve in the blocks of the reference's OpenMP threads: the host restatement of the thread decomposition
against indice_boucle_lu_ of the compiled reference, and the CPU restatement of the block-wise solve against invlu_ of
the compiled reference called once per thread block (Fast/FastS/FastS/Compute/lhs.cpp:129-161)."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal
from oracle import oracle as O
from oracle import ref_harness as R
from common import prepare, zone_arrays, wall_distance, SINGLE

FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}


def _pint(ni, nj, nk):
    p = np.zeros(64, dtype=np.int32)
    p[FastC.IJKV:FastC.IJKV + 3] = (ni, nj, nk)
    return p


def test_blocks_tile_the_zone():
    for size in ((256, 256, 256), (100, 37, 64), (1024, 256, 1), (33, 21, 45)):
        for n in (2, 3, 4, 6, 8, 12, 16, 24, 48, 64, 128):
            b = FastC.lu_thread_blocks(_pint(*size), n)
            assert len(b) <= n
            vol = ((b[:, 1] - b[:, 0] + 1) * (b[:, 3] - b[:, 2] + 1) * (b[:, 5] - b[:, 4] + 1)).sum()
            assert vol == size[0] * size[1] * size[2], (size, n)
            seen = np.zeros(size, dtype=np.int8)
            for q in b:
                seen[q[0] - 1:q[1], q[2] - 1:q[3], q[4] - 1:q[5]] += 1
            assert (seen == 1).all(), (size, n)
    # the benchmark block on 64 threads: 4 x 16 blocks of 256 x 64 x 16 cells
    b = FastC.lu_thread_blocks(_pint(256, 256, 256), 64)
    assert len(b) == 64 and tuple(b[0]) == (1, 256, 1, 64, 1, 16) and tuple(b[-1]) == (1, 256, 193, 256, 241, 256)


@pytest.mark.ref
@pytest.mark.skipif(not R.available(), reason="compiled reference not built")
def test_thread_blocks_equal_indice_boucle_lu():
    """Every thread's range for a sweep of zone sizes and thread counts (three- and two-dimensional zones, sizes that
    leave remainders, more threads than slices of 10 cells)."""
    L = R.lib()
    import ctypes
    rng = np.random.RandomState(5)
    sizes = [(256, 256, 256), (64, 64, 64), (100, 37, 64), (1024, 256, 1), (384, 96, 1), (33, 21, 45), (12, 24, 22),
             (25, 25, 9), (9, 95, 130), (300, 12, 1)]
    sizes += [tuple(int(v) for v in rng.randint(5, 140, size=3)) for _ in range(25)]
    sizes += [(int(rng.randint(20, 400)), int(rng.randint(10, 200)), 1) for _ in range(10)]
    checked = 0
    for size in sizes:
        ind_dm = np.array([1, size[0], 1, size[1], 1, size[2]], dtype=np.int32)
        for n in list(range(1, 33)) + [36, 40, 48, 56, 64, 96, 128]:
            ref = []
            topo = np.ones(3, dtype=np.int32)
            for ith in range(1, n + 1):
                ind = np.zeros(6, dtype=np.int32)
                L.indice_boucle_lu_(R._i(0), R._i(ith), R._i(n), R._i(10), R._p(ind_dm), R._p(topo), R._p(ind))
                if ind[1] >= ind[0] and ind[3] >= ind[2] and ind[5] >= ind[4]:
                    ref.append(ind)
            got = FastC.lu_thread_blocks(_pint(*size), n)
            assert np.array_equal(got, np.array(ref, dtype=np.int32)), (size, n)
            checked += 1
    assert checked > 1500


def _pair(t):
    metrics = prepare(t)
    z = Internal.getZones(t)[0]
    pint, preal, xyz = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z), metrics[0][0], wall_distance(z))
    rz = R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], FastC.level_buffer(z), metrics[0][0], wall_distance(z))
    return O.OracleSolver([oz], []), R.RefSolver([rz], [])


@pytest.mark.ref
@pytest.mark.skipif(not R.available(), reason="compiled reference not built")
@pytest.mark.parametrize("n,scheme,model,perturb,bcs,threads,nb_relax", [
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 4, 1),
    ((12, 24, 33), 'senseur', 'NSLaminar', 0., None, 6, 1),
    ((12, 23, 21), 'roe', 'Euler', 0.15, None, 4, 1),
    ((12, 31, 10), 'senseur', 'NSLaminar', 0.15, None, 3, 1),
    # walls: a block that touches the wall face takes its ghost layer
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 4, 1),
    ((11, 22, 23), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 4, 1),
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., FAR, 2, 1),
    # Spalart-Allmaras (6 equations), Cartesian and curvilinear
    ((12, 24, 22), 'ausmpred', 'NSTurbulent', 0., {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 4, 1),
    ((12, 22, 21), 'roe', 'NSTurbulent', 0.15, None, 4, 1),
    # two-dimensional zones: cut along j and i
    ((34, 22), 'ausmpred', 'NSLaminar', 0.15, None, 4, 1),
    ((45, 23), 'senseur', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 6, 1),
    # LU-SSOR relaxations, every thread on work arrays of its own window
    ((12, 24, 22), 'ausmpred', 'NSLaminar', 0., None, 4, 2),
    ((12, 23, 21), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 4, 3),
    ((12, 24, 22), 'roe', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 4, 2),
    ((45, 23), 'senseur', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 6, 2),
])
def test_block_lusgs_bit_exact(n, scheme, model, perturb, bcs, threads, nb_relax):
    make = cases.tgv if len(n) == 3 else cases.tgv2d
    SINGLE = dict(globals()['SINGLE'], nb_relax=nb_relax)
    t = make(n, scheme, model=model, extra_numz=SINGLE, extra_numb={'ss_iteration': 3}, bcs=bcs,
             temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    so, sr = _pair(t)
    blocks = FastC.lu_thread_blocks(so.zones[0].pint, threads)
    assert len(blocks) == threads
    so.zones[0].lu_blocks = blocks
    sr.lu_threads = threads
    assert np.array_equal(sr.lu_thread_blocks(0, sr.zones[0], threads), blocks)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-4
    # the blocks matter: one sweep of the zone gives another field
    t1 = make(n, scheme, model=model, extra_numz=SINGLE, extra_numb={'ss_iteration': 3}, bcs=bcs,
              temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    s1, _ = _pair(t1)
    s1.fill_ghostcells()
    for a in s1.zones:
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro
    for it in range(1, 5):
        s1.step()
    assert not np.array_equal(s1.zones[0].ro, so.zones[0].ro)
