"""GPU parity of the Spalart-Allmaras model (NSTurbulent, 6 equations) on Cartesian zones against the CPU
restatement, itself bit-exact against the compiled reference (test_oracle_vs_ref.py).  The destruction
term takes a 1/6 power (fw function, SA/spsource_SA.for:75): the device pow and the host pow differ in
the last bit, so fields are compared to 1e-12 relative (the tolerance of the path is 1e-11 after 100
iterations); every other operation keeps the reference's order."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal, _capi
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare, wall_distance

pytestmark = pytest.mark.gpu

CASES = [
    ((14, 12, 10), 'ausmpred', None, 6),
    ((12, 10, 9), 'senseur', None, 6),
    ((70, 17, 37), 'senseur', None, 3),
    ((12, 9, 10), 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 6),
    ((10, 9, 8), 'senseur', {'jmin': 'BCWall', 'kmax': 'BCFarfield', 'imin': 'BCExtrapolate', 'imax': 'BCWallInviscid'}, 5),
    ((12, 10, 9), 'roe', None, 6),
    ((10, 9, 8), 'senseur', {'imin': 'BCInflow', 'imax': 'BCExtrapolate', 'jmax': 'BCInflowSupersonic'}, 5),
    ((10, 9, 8), 'ausmpred', {'imin': 'BCExtrapolateRANS', 'kmax': 'BCExtrapolateRANS', 'jmin': 'BCWall'}, 5),
]


CASES.append(((10, 9, 8), 'ausmpred', 'own_state', 5))


@pytest.mark.parametrize("n,scheme,bcs,nit", CASES)
def test_sa_matches_oracle(n, scheme, bcs, nit):
    if bcs == 'own_state':
        from common import own_state_bcs
        bcs = own_state_bcs()
    t = cases.tgv(n, scheme, model='NSTurbulent', extra_numz=SINGLE, bcs=bcs)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[36]) == 6 and int(pint[27]) == 3
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)        # 6-variable ghost fill: bit-exact
    ro0 = oz.ro.copy()
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]), neq=6)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    d = np.abs(got - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(got).all() and d[5] > 1e-7


IMPLICIT_CASES = [
    ((14, 12, 10), 'ausmpred', None, False, 4),
    ((12, 10, 9), 'senseur', None, False, 4),
    ((12, 10, 9), 'roe', None, False, 4),
    ((40, 11, 21), 'ausmpred', None, False, 3),                     # several 8-cell units per line, ragged last unit
    ((12, 9, 10), 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, False, 4),
    ((10, 9, 8), 'ausmpred', {'jmin': 'BCWall', 'kmax': 'BCFarfield', 'imin': 'BCExtrapolate', 'imax': 'BCWallInviscid'}, True, 4),
]


@pytest.mark.parametrize("n,scheme,bcs,local,nit", IMPLICIT_CASES)
def test_sa_implicit_matches_oracle(n, scheme, bcs, local, nit):
    """6-equation Newton residual, coe(:,6), LU-SGS sweeps with the scalar 6th equation and the clipped Newton
    update; BDF1 at the first step then BDF2 (or local time step: BDF1 throughout)."""
    extra = dict(SINGLE)
    if local:
        extra.update({'time_step_nature': 'local', 'cfl': 4.})
    t = cases.tgv(n, scheme, model='NSTurbulent', extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[36]) == 6 and int(pint[29]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]), neq=6)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    d = np.abs(got - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(got).all() and d[5] > 1e-7 and d[1] > 1e-4


GENERAL_CASES = [
    (3, (14, 12, 10), 'ausmpred', None, False),
    (3, (12, 10, 9), 'senseur', None, False),
    (3, (12, 10, 9), 'roe', None, False),
    (3, (40, 11, 12), 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    (3, (14, 12, 10), 'ausmpred', None, True),
    (3, (12, 10, 9), 'roe', None, True),
    (3, (12, 10, 9), 'senseur', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    (2, (14, 12), 'ausmpred', None, False),
    (2, (12, 10), 'senseur', None, False),
    (2, (12, 10), 'roe', None, False),
    (2, (70, 17), 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    (2, (14, 12), 'ausmpred', None, True),
    (2, (12, 10), 'roe', None, True),
    (2, (40, 11), 'senseur', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    (2, (40, 11), 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 'local'),
]


@pytest.mark.parametrize("dim,n,scheme,bcs,implicit", GENERAL_CASES)
def test_sa_general_mesh_matches_oracle(dim, n, scheme, bcs, implicit):
    """NSTurbulent on curvilinear (zone type 0) and two-dimensional (type 3) zones, explicit and implicit."""
    extra = dict(SINGLE)
    kw = {}
    if implicit:
        kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})
        if implicit == 'local':
            extra.update({'time_step_nature': 'local', 'cfl': 4.})
    if dim == 3:
        t = cases.tgv(n, scheme, model='NSTurbulent', extra_numz=extra, bcs=bcs, perturb=0.15, **kw)
    else:
        t = cases.tgv2d(n, scheme, model='NSTurbulent', extra_numz=extra, bcs=bcs, perturb=0.15, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[36]) == 6 and int(pint[25]) == (0 if dim == 3 else 3)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z))
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
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


def test_sa_blocks_with_joins():
    blocks, nn = (2, 1, 2), (16, 9, 12)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    kw = dict(model='NSTurbulent', blocks=blocks, extra_numz=SINGLE, bcs={'jmin': 'BCWall'})
    to = cases.tgv(nn, 'senseur', **kw)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0], wall_distance(z)))
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
            e = rel_err(FastC.level_buffer(z), oz.ro, oz.ndimdx, neq=6)
            assert e.max() < 1e-12, "step %d zone %s: %s" % (it, z[0], e)


def test_sa_options_outside_the_device_path_are_refused():
    t = cases.tgv(8, 'senseur', model='NSTurbulent')
    for zz in Internal.getZones(t):      # no wall distance
        sol = Internal.getNodeFromName1(zz, 'FlowSolution#Centers')
        sol[2] = [n for n in sol[2] if n[0] != 'TurbulentDistance']
    with pytest.raises(Exception):
        FastS.warmup(t, None)
