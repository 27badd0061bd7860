"""Long runs against the CPU restatement (k-slabs on OpenMP threads, bit-identical to its one-thread sweep:
test_oracle_threads.py): 100 time steps at 64^3, 10 time steps at the benchmark size 256^3, 100 steps of the
Spalart-Allmaras model, and a restart from a tree that carries the M1 time level."""
import os
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal, _capi
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, wall_distance

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))


def _pair(t, dist=False):
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    p, pr, _ = zone_arrays(z)
    oz = O.OracleZone(p, pr, FastC.level_buffer(z), metrics[0][0], wall_distance(z) if dist else None)
    oz.threads = THREADS
    return t, metrics, z, oz


@pytest.mark.parametrize("n,scheme,nit", [(64, 'senseur', 100), (256, 'senseur', 10), (64, 'ausmpred', 30)])
def test_many_steps_bit_exact(n, scheme, nit):
    os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
    try:
        t, metrics, z, oz = _pair(cases.tgv(n, scheme, extra_numz=SINGLE))
        for it in range(1, nit + 1):
            FastS._compute(t, metrics, it, None)
            oz.step()
            if it % 10 == 0 or it == nit:
                FastS.sync_to_host(t)
                got = FastC.level_buffer(z)
                same = np.array_equal(got, oz.ro)
                assert same, "step %d: max abs diff %g" % (it, np.abs(got - oz.ro).max())
        assert np.isfinite(got).all()
    finally:
        os.environ.pop('FASTS_HIP_LAZY_SYNC', None)


def test_config1_vortex_64_roe_hundred_steps_bit_exact():
    """The first benchmark configuration as it is named: single Cartesian block of 64^3 cells, inviscid isentropic
    vortex, explicit RK3 + roe (o3 slopes), six periodic faces, 100 time steps: the field after every tenth step
    equals the CPU restatement's bit for bit."""
    os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
    try:
        t, metrics, z, oz = _pair(cases.vortex(64, 'roe', extra_numz=SINGLE))
        p = zone_arrays(z)[0]
        assert (int(p[20]), int(p[21]), int(p[22])) == (64, 64, 64) and int(p[33]) == 5 and int(p[27]) == 1 and int(p[64]) == 2
        for it in range(1, 101):
            FastS._compute(t, metrics, it, None)
            oz.step()
            if it % 10 == 0:
                FastS.sync_to_host(t)
                got = FastC.level_buffer(z)
                assert np.array_equal(got, oz.ro), "step %d: max abs diff %g" % (it, np.abs(got - oz.ro).max())
        assert np.isfinite(got).all()
    finally:
        os.environ.pop('FASTS_HIP_LAZY_SYNC', None)


@pytest.mark.parametrize("scheme,bcs", [('senseur', None), ('roe_min', {'jmin': 'BCWall', 'jmax': 'BCOutflow'})])
def test_curvilinear_hundred_steps_bit_exact(scheme, bcs):
    """100 steps on a curvilinear 48 x 40 x 36 zone (several patches and k chunks of the stage kernel that keeps the state
    planes in LDS): bit for bit, the 0/1 sensor of senseur included."""
    os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
    try:
        t, metrics, z, oz = _pair(cases.tgv((48, 40, 36), scheme, extra_numz=SINGLE, perturb=0.2, bcs=bcs))
        oz.fill_ghostcells()
        for it in range(1, 101):
            FastS._compute(t, metrics, it, None)
            oz.step()
            if it % 25 == 0:
                FastS.sync_to_host(t)
                got = FastC.level_buffer(z)
                assert np.array_equal(got, oz.ro), "step %d: max abs diff %g" % (it, np.abs(got - oz.ro).max())
        assert np.isfinite(got).all()
    finally:
        os.environ.pop('FASTS_HIP_LAZY_SYNC', None)


def test_sa_hundred_steps():
    """The device pow of the destruction term differs from the host pow in the last bit: 1e-11 after 100 steps."""
    t, metrics, z, oz = _pair(cases.tgv((24, 20, 16), 'senseur', model='NSTurbulent', extra_numz=SINGLE,
                                        bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'}), dist=True)
    oz.fill_ghostcells()
    nd = int(oz.pint[41])
    for it in range(1, 101):
        FastS._compute(t, metrics, it, None)
        oz.step()
    e = rel_err(FastC.level_buffer(z), oz.ro, nd, neq=6)
    print("SA, 100 steps: largest relative error per equation %s; margin to 1e-11: x%.0f" % (e, 1e-11 / max(e.max(), 1e-300)))
    assert e.max() < 1e-11, e


IMPLICIT = {'time_step': 0.02, 'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': 3}}


def test_restart_with_stored_m1_level_continues_in_bdf2():
    """A tree that carries Density_M1 ... starts with FIRST_IT = 1 (Fast/FastC/FastC/PyTree.py:2195-2222): the second
    order scheme from the first step on.  Two runs of two steps chained through such a tree equal one run of four."""
    n = (12, 10, 9)
    t, metrics, z, oz = _pair(cases.tgv(n, 'senseur', extra_numz=SINGLE, **IMPLICIT))
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
    whole = FastC.level_buffer(z).copy()
    assert np.array_equal(whole, oz.ro)

    t1 = cases.tgv(n, 'senseur', extra_numz=SINGLE, **IMPLICIT)
    (t1, _, m1) = FastS.warmup(t1, None)
    assert FastC.FIRST_IT == 0
    for it in range(1, 3):
        FastS._compute(t1, m1, it, None)
    FastS.sync_to_host(t1)
    z1 = Internal.getZones(t1)[0]
    sol1 = Internal.getNodeFromName1(z1, 'FlowSolution#Centers')
    saved = {v + s: np.array(Internal.getNodeFromName1(sol1, v + s)[1], copy=True, order='F')
             for v in FastC.zone_vars(z1) for s in ('', '_M1')}

    t2 = cases.tgv(n, 'senseur', extra_numz=SINGLE, **IMPLICIT)
    sol2 = Internal.getNodeFromName1(Internal.getZones(t2)[0], 'FlowSolution#Centers')
    for name, a in saved.items():
        node = Internal.getNodeFromName1(sol2, name)
        if node is None:
            Internal.createChild(sol2, name, 'DataArray_t', a)
        else:
            node[1] = a
    (t2, _, m2) = FastS.warmup(t2, None)
    assert FastC.FIRST_IT == 1
    for it in range(3, 5):
        FastS._compute(t2, m2, it, None)
    got = FastC.level_buffer(Internal.getZones(t2)[0])
    assert np.array_equal(got, whole), "max abs diff %g" % np.abs(got - whole).max()
