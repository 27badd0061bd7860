"""GPU parity on two-dimensional zones (zone type 3: one cell along k, surface vectors with two components)
against the CPU restatement, itself bit-exact against the compiled reference (test_oracle_vs_ref.py)."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, wall_distance

pytestmark = pytest.mark.gpu

CASES = [
    ((14, 12), 'ausmpred', 'NSLaminar', 0.15, None, False),
    ((12, 10), 'senseur', 'NSLaminar', 0.15, None, False),
    ((12, 9), 'roe', 'Euler', 0.15, None, False),
    ((70, 17), 'senseur', 'NSLaminar', 0.1, None, False),                   # several patches, ragged
    ((12, 10), 'senseur', 'Euler', 0., None, False),
    ((12, 9), 'ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    ((12, 9), 'senseur', 'Euler', 0.15, {'jmin': 'BCWallInviscid', 'imax': 'BCExtrapolate', 'imin': 'BCFarfield'}, False),
    ((14, 12), 'ausmpred', 'NSLaminar', 0.15, None, True),
    ((12, 9), 'roe', 'NSLaminar', 0.15, None, True),
    ((40, 11), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    ((12, 9), 'ausmpred', 'Euler', 0., {'jmin': 'BCWallInviscid', 'jmax': 'BCFarfield'}, 'local'),
]


@pytest.mark.parametrize("n,scheme,model,perturb,bcs,implicit", CASES)
def test_two_dimensional_zone_bit_exact(n, scheme, model, perturb, bcs, implicit):
    extra = dict(SINGLE)
    kw = {}
    if implicit:
        kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})
        if implicit == 'local':
            extra.update({'time_step_nature': 'local', 'cfl': 4.})
    t = cases.tgv2d(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[25]) == 3 and int(pint[4]) == 0
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-5
    assert np.all(got.reshape(5, -1)[3] == 0.)        # no velocity along z
