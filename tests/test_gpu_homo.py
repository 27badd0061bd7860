"""GPU parity on k-homogeneous zones (zone type 1, an extruded mesh: metrics of one k-plane, ti / tj without a z
component, tk with a z component only).  The device runs them through the curvilinear kernels with zero-padded metrics
that have no stride along k, and takes the z derivatives of the viscous flux and of the Spalart-Allmaras source in
the factored form of the reference's 3dhomo fragments.  Checked bit for bit against the CPU restatement, itself
bit-exact against the compiled 3dhomo fragments (test_oracle_vs_ref.py::test_homogeneous_zone_bit_exact)."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare, wall_distance

pytestmark = pytest.mark.gpu

FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}
IMPL = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})

CASES = [
    # n, scheme, model, bcs, options, steps
    ((14, 12, 10), 'senseur', 'NSLaminar', None, {}, 6),
    ((12, 10, 9), 'ausmpred', 'NSLaminar', None, {}, 5),
    ((9, 8, 7), 'roe', 'NSLaminar', None, {}, 5),
    ((10, 9, 12), 'roe', 'Euler', None, {}, 4),
    ((12, 9, 10), 'senseur', 'Euler', None, {}, 4),
    ((70, 17, 37), 'senseur', 'NSLaminar', None, {}, 3),       # several patches in i and j, several k chunks
    ((12, 10, 9), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, {}, 5),
    ((10, 9, 8), 'roe', 'NSLaminar', {'jmin': 'BCWall', 'kmax': 'BCWall'}, {}, 5),      # wall-flux correction
    ((10, 9, 8), 'ausmpred', 'Euler', {'kmin': 'BCWallInviscid', 'imax': 'BCSymmetryPlane'}, {}, 5),
    ((9, 8, 10), 'senseur', 'NSLaminar', FAR, {}, 5),
    ((12, 10, 9), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'kmin': 'BCFarfield'}, IMPL, 4),
    ((10, 12, 9), 'senseur', 'NSLaminar', None, IMPL, 4),
    ((12, 10, 9), 'roe', 'NSLaminar', {'jmin': 'BCWall'}, dict(IMPL, extra_numz={'nb_relax': 2}), 4),     # LU-SSOR
]


@pytest.mark.parametrize("n,scheme,model,bcs,opts,nit", CASES)
def test_homogeneous_zone_bit_exact_vs_oracle(n, scheme, model, bcs, opts, nit):
    opts = dict(opts)
    extra = dict(SINGLE, **opts.pop('extra_numz', {}))
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=0.2, homo=True, **opts)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[25]) == 1 and int(pint[37]) == 2 and int(pint[38]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    ro0 = oz.ro.copy()
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4


@pytest.mark.parametrize("scheme,bcs,implicit", [
    ('senseur', None, False), ('roe', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    ('ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
])
def test_homogeneous_zone_spalart_allmaras(scheme, bcs, implicit):
    """NSTurbulent: source term and nu_tilde diffusion with the factored z derivatives (device pow: 1e-12)."""
    kw = dict(IMPL) if implicit else {}
    t = cases.tgv((12, 10, 9), scheme, model='NSTurbulent', extra_numz=SINGLE, bcs=bcs, perturb=0.2, homo=True, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[36]) == 6 and int(pint[25]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]), neq=6)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    d = np.abs(got - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(got).all() and d[5] > 1e-7 and d[1] > 1e-4


def test_homogeneous_blocks_with_joins():
    blocks, nn = (2, 1, 2), (16, 9, 12)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    kw = dict(model='NSLaminar', blocks=blocks, extra_numz=SINGLE, perturb=0.2, homo=True)
    to = cases.tgv(nn, 'senseur', **kw)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
        assert int(p[25]) == 1
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in joins])
    s.fill_ghostcells()
    t = cases.tgv(nn, 'senseur', **kw)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for it in range(1, 4):
        FastS._compute(t, metrics, it, tc)
        s.step()
        for z, oz in zip(zones, s.zones):
            assert np.array_equal(FastC.level_buffer(z), oz.ro), "step %d zone %s" % (it, z[0])


def test_homogeneous_cfl_diagnostic():
    """CFL_minmaxmoy of a verification step (cpcfl3.for:142-187): max and min bit for bit, the mean to 1e-13."""
    t = cases.tgv((34, 9, 8), 'ausmpred', extra_numz=SINGLE, perturb=0.2, homo=True)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[25]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    want = oz.cfl()
    FastS._compute(t, metrics, 1, None)
    FastS.display_temporal_criteria(t, metrics, 1, format='none')
    got = Internal.getNodeFromName1(Internal.getNodeFromName1(z, '.Solver#ownData'), 'CFL_minmaxmoy')[1]
    assert got[0] == want[0] and got[1] == want[1]
    np.testing.assert_allclose(got[2], want[2], rtol=1e-13)
    assert want[0] > want[1] > 0.
