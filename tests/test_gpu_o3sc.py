"""slope = 'o3sc' on the device (shock-sensor kernel + the one-thread-per-cell stage kernels) against the CPU restatement,
which is pinned bit for bit on the compiled reference's flu*_o3sc_* variants (tests/test_oracle_vs_ref.py).  The sensor
takes a tanh, and the device's differs from the host's in the last bit: fields to 1e-12 relative instead of equality."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal, _capi
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE

pytestmark = pytest.mark.gpu

IMPL = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})


def _front(t):
    """A compression front across the zone: u falls by 4 over a few cells along i (the sensor then departs from 1)."""
    for z in Internal.getZones(t):
        u = Internal.getNodeFromName2(z, 'VelocityX')[1]
        ni = u.shape[0]
        u -= 2. * np.tanh((np.arange(ni) - ni / 2.) / 1.5)[:, None, None]


@pytest.mark.parametrize("dim,n,scheme,model,bcs,perturb,kw", [
    (3, (20, 12, 10), 'ausmpred', 'NSLaminar', None, 0., {}),
    (3, (70, 9, 8), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., {}),
    (3, (12, 10, 9), 'roe', 'Euler', None, 0., {}),
    (3, (12, 10, 9), 'senseur', 'Euler', None, 0.15, {}),
    (3, (40, 10, 9), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'kmax': 'BCFarfield'}, 0.15, {}),
    (3, (12, 10, 9), 'roe', 'NSLaminar', None, 0.15, {}),
    (3, (12, 10, 9), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall'}, 0.2, {'homo': True}),
    (2, (24, 16), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, {}),
    (3, (12, 10, 9), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall'}, 0.15, IMPL),
    (3, (16, 10, 9), 'senseur', 'NSLaminar', None, 0., IMPL),
])
def test_o3sc_vs_oracle(dim, n, scheme, model, bcs, perturb, kw):
    extra = dict(SINGLE, slope='o3sc')
    if dim == 2:
        t = cases.tgv2d(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    _front(t)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[64]) == 6
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    nd = int(pint[41])
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, nd)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    assert np.isfinite(got).all()
    assert oz.wig[-3 * nd:].min() < 0.999, "the sensor never departed from 1"
    FastC.HOOK['ctx'].close()


def test_o3sc_refusals():
    """Spalart-Allmaras and roe + walls with this slope are refused with a message."""
    for scheme, model, bcs in (('ausmpred', 'NSTurbulent', {'jmin': 'BCWall'}), ('roe', 'NSLaminar', {'jmin': 'BCWall'})):
        t = cases.tgv((10, 9, 8), scheme, model=model, extra_numz=dict(SINGLE, slope='o3sc'), bcs=bcs)
        with pytest.raises(_capi.FastsHipError, match="o3sc"):
            FastS.warmup(t, None)
