"""GPU parity of roe + walls: the flux through a wall face is replaced by the wall flux (BC/bfl3.for,
bflwall.for, bflwallslip.for).  The CPU restatement is bit-exact against the compiled reference
(test_oracle_vs_ref.py::test_roe_wall_flux_correction_bit_exact)."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare, wall_distance

pytestmark = pytest.mark.gpu

CASES = [
    ('Euler', 0., 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, False),
    ('NSLaminar', 0., 3, (70, 9, 37), {'jmin': 'BCWall', 'kmax': 'BCWallViscous', 'imin': 'BCSymmetryPlane'}, False),
    ('NSLaminar', 0.15, 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid', 'imax': 'BCFarfield'}, False),
    ('NSTurbulent', 0., 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, False),
    ('NSTurbulent', 0.15, 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, False),
    ('NSLaminar', 0.15, 2, (40, 11), {'jmin': 'BCWall', 'jmax': 'BCFarfield', 'imin': 'BCWallInviscid'}, False),
    ('NSLaminar', 0., 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, True),
    ('NSLaminar', 0.15, 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, True),
    ('NSTurbulent', 0., 3, (12, 9, 10), {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    ('NSTurbulent', 0.15, 2, (40, 11), {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
]


@pytest.mark.parametrize("model,perturb,dim,n,bcs,implicit", CASES)
def test_roe_walls_match_oracle(model, perturb, dim, n, bcs, implicit):
    kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3}) if implicit else {}
    if dim == 3:
        t = cases.tgv(n, 'roe', model=model, extra_numz=SINGLE, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv2d(n, 'roe', model=model, extra_numz=SINGLE, bcs=bcs, perturb=perturb, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    tol = 1e-12 if neq == 6 else 0.
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        e = rel_err(FastC.level_buffer(z), oz.ro, int(pint[41]), neq=neq)
        assert e.max() <= tol, "step %d: rel err %s" % (it, e)
    assert np.isfinite(oz.ro).all() and np.abs(oz.ro - ro0).max() > 1e-5


def test_naca0012_roe():
    """The steady-RANS set-up of the reference's own SA case (roe, implicit, local time step) on the C-mesh."""
    nf, nw, nj = 16, 6, 10
    kw = dict(nfoil=nf, nwake=nw, nj=nj, scheme='roe', extra_numz=SINGLE)
    rp, dp = connect.cmesh_wake_join(2 * nf + 2 * nw, nj, nw)
    to = cases.naca_cmesh(**kw)
    m = prepare(to)
    zo = Internal.getZones(to)[0]
    p, pr, _ = zone_arrays(zo)
    oz = O.OracleZone(p, pr, FastC.level_buffer(zo), m[0][0], wall_distance(zo))
    s = O.OracleSolver([oz], [(0, rp, 0, dp)])
    s.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    t = cases.naca_cmesh(**kw)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, [(0, 0, rp, dp)], [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    z = Internal.getZones(t)[0]
    for it in range(1, 9):
        FastS._compute(t, metrics, it, tc)
        s.step()
        e = rel_err(FastC.level_buffer(z), oz.ro, oz.ndimdx, neq=6)
        assert e.max() < 1e-12, "step %d: %s" % (it, e)
    assert np.isfinite(oz.ro).all()
