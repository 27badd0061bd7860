"""Full-size checks (256^3, the benchmark configuration): one complete time step against the CPU
restatement, and size-independent properties of the scheme on a periodic Cartesian box --
translation invariance of the inviscid scheme (bit-exact: every cell runs the same arithmetic
wherever the tile and chunk boundaries fall; the viscous cross derivatives are excluded because
next to the box boundary they read edge ghost cells, which the reference fills with averages) and conservation of mass, momentum and total energy."""
import os
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays

pytestmark = pytest.mark.gpu

N = 256
SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def _interior(z, buf=None):
    p, _, _ = zone_arrays(z)
    ni, nj, nk = int(p[0]), int(p[1]), int(p[2])
    nd = int(p[FastC.NDIMDX])
    b = FastC.level_buffer(z) if buf is None else buf
    return b.reshape(5, nd)[:, :ni * nj * nk].reshape(5, nk, nj, ni)[:, 2:-2, 2:-2, 2:-2]


def test_one_full_size_step_bit_exact_vs_oracle():
    t = cases.tgv(N, 'senseur', extra_numz=SINGLE)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    p, pr, _ = zone_arrays(z)
    oz = O.OracleZone(p, pr, FastC.level_buffer(z), metrics[0][0])
    oz.threads = max(1, min(16, os.cpu_count() or 1))
    FastS._compute(t, metrics, 1, None)
    oz.step()
    got = FastC.level_buffer(z)
    same = np.array_equal(got, oz.ro)
    assert same, "max abs diff %g" % np.abs(got - oz.ro).max()


@pytest.mark.parametrize("scheme", ["senseur", "ausmpred"])
def test_translation_invariance_full_size(scheme):
    shift = (11, 5, 37)          # cells along k, j, i: moves every tile and chunk boundary
    nit = 2
    t = cases.tgv(N, scheme, model='Euler')
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    start = _interior(z).copy()
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
    ref = np.roll(_interior(z), shift, axis=(1, 2, 3))

    t2 = cases.tgv(N, scheme, model='Euler')
    (t2, tc2, metrics2) = FastS.warmup(t2, None)
    z2 = Internal.getZones(t2)[0]
    _interior(z2)[...] = np.roll(start, shift, axis=(1, 2, 3))
    FastS.sync_to_device(t2)
    FastS._fillGhostcells([z2], None, metrics2, 1, ['Density'], 1, 0, None)
    for it in range(1, nit + 1):
        FastS._compute(t2, metrics2, it, None)
    got = _interior(z2)
    same = np.array_equal(got, ref)
    assert same, "max abs diff %g" % np.abs(got - ref).max()


def test_conservation_full_size():
    t = cases.tgv(N, 'senseur')
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    _, pr, _ = zone_arrays(z)
    cv = pr[FastC.CVINF]

    def totals():
        q = _interior(z).astype(np.longdouble)
        ro = q[0]
        return np.array([ro.sum(), (ro * q[1]).sum(), (ro * q[2]).sum(), (ro * q[3]).sum(),
                         (ro * (cv * q[4] + 0.5 * (q[1] ** 2 + q[2] ** 2 + q[3] ** 2))).sum()])

    a = totals()
    for it in range(1, 4):
        FastS._compute(t, metrics, it, None)
    b = totals()
    ncell = float(N) ** 3
    assert abs(b[0] - a[0]) / a[0] < 1e-13
    assert abs(b[4] - a[4]) / a[4] < 1e-13
    # momentum sums to zero for the Taylor-Green field: compare with the scale rho*U0*ncell
    assert np.abs(b[1:4] - a[1:4]).max() / ncell < 1e-13


def test_implicit_step_full_size_one_launch_sweeps_equal_per_hyperplane_sweeps(monkeypatch):
    """256^3, two implicit time steps of three Newton sub-iterations: the LU-SGS sweeps as one launch each (8 192 pencils,
    768 in work at a time) give the fields of the sweeps with one launch per hyperplane, bit for bit."""
    def run(flow):
        monkeypatch.setenv('FASTS_HIP_LU_FLOW', str(flow))
        t = cases.tgv(N, 'senseur', extra_numz=SINGLE, temporal_scheme='implicit', time_step=0.003,
                      extra_numb={'ss_iteration': 3})
        (t, tc, metrics) = FastS.warmup(t, None)
        z = Internal.getZones(t)[0]
        for it in range(1, 3):
            FastS._compute(t, metrics, it, None)
        out = FastC.level_buffer(z).copy()
        FastC.HOOK['ctx'].close()
        return out
    ref = run(0)
    got = run(1)
    assert np.isfinite(ref).all()
    assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()
