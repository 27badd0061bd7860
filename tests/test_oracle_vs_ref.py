"""CPU restatement and host model against the compiled reference itself (oracle/_ref), on
cases beyond the committed golden vectors.  Skipped where the reference library is absent."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
from oracle import oracle as O
from oracle import ref_harness as R
from common import prepare, zone_arrays, make_case, wall_distance

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not R.available(), reason="compiled reference not built")]


def _pair(t, blocks=None, nv=None):
    metrics = prepare(t)
    zones = Internal.getZones(t)
    oz, rz = [], []
    for z, m in zip(zones, metrics):
        pint, preal, xyz = zone_arrays(z)
        oz.append(O.OracleZone(pint, preal, FastC.level_buffer(z), m[0], wall_distance(z)))
        rz.append(R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], FastC.level_buffer(z), m[0], wall_distance(z)))
    joins = []
    if blocks is not None:
        joins = [(br, r, bd, d) for br, bd, r, d in connect.periodic_block_joins(blocks, nv)]
    return O.OracleSolver(oz, joins), R.RefSolver(rz, joins)


@pytest.mark.parametrize("kind,n,scheme,model,nit", [
    ('tgv', 14, 'senseur', 'NSLaminar', 6),
    ('tgv', (9, 7, 5), 'ausmpred', 'NSLaminar', 4),
    ('tgv', (6, 11, 8), 'roe', 'NSLaminar', 4),
    ('tgv', 9, 'ausmpred', 'Euler', 4),
    ('vortex', 10, 'roe', None, 6),
])
def test_single_zone_bit_exact(kind, n, scheme, model, nit):
    so, sr = _pair(make_case(kind, n, scheme, model))
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for it in range(1, nit + 1):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
        np.testing.assert_allclose(so.zones[0].residuals()[0], sr.residuals()[0][0], rtol=1e-12)


FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}


@pytest.mark.parametrize("model,scheme,bcs", [
    ('NSLaminar', 'senseur', {'jmin': 'BCWall', 'jmax': 'BCWall'}),
    ('NSLaminar', 'ausmpred', {'imin': 'BCWallViscous', 'kmax': 'BCWallViscous'}),
    ('Euler', 'ausmpred', {'kmin': 'BCWallInviscid', 'kmax': 'BCWallInviscid', 'imax': 'BCSymmetryPlane'}),
    ('Euler', 'senseur', {'jmin': 'BCWall', 'imin': 'BCWall'}),          # tag 4 is a slip wall for Euler
    ('NSLaminar', 'senseur', FAR),
    ('Euler', 'roe', {'imin': 'BCExtrapolate', 'imax': 'BCExtrapolate', 'jmax': 'BCFarfield', 'kmin': 'BCExtrapolate'}),
    ('NSLaminar', 'ausmpred', {'imin': 'BCInflow', 'imax': 'BCFarfield', 'jmin': 'BCWall', 'kmax': 'BCInflowSupersonic'}),
    ('NSTurbulent', 'senseur', {'imin': 'BCInflow', 'imax': 'BCExtrapolate', 'jmax': 'BCInflowSupersonic'}),
    ('NSTurbulent', 'ausmpred', {'imin': 'BCExtrapolateRANS', 'kmax': 'BCExtrapolateRANS', 'jmin': 'BCWall'}),
    ('NSLaminar', 'ausmpred', {'imax': 'BCExtrapolateRANS'}),      # 5 equations: the routine changes nothing
    ('NSLaminar', 'senseur', 'own_state'),
    ('NSTurbulent', 'ausmpred', 'own_state'),
])
def test_boundary_conditions_bit_exact(model, scheme, bcs):
    """Ghost fill of walls, far field and extrapolation against bvbs_wall_viscous_adia /
    bvbs_wall_inviscid / bvbs_farfield / bvbs_extrapolate of the compiled reference."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if bcs == 'own_state':
        from common import own_state_bcs
        bcs = own_state_bcs()
    t = cases.tgv((9, 8, 7), scheme, model=model, extra_numz=extra, bcs=bcs)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all()


@pytest.mark.parametrize("model,scheme,perturb,temporal", [
    ('NSLaminar', 'ausmpred', 0., 'explicit'),
    ('Euler', 'senseur', 0., 'explicit'),
    ('NSLaminar', 'roe', 0.15, 'explicit'),
    ('NSTurbulent', 'ausmpred', 0.15, 'explicit'),
    ('NSLaminar', 'senseur', 0.15, 'implicit'),
    ('NSTurbulent', 'ausmpred', 0., 'implicit'),
])
def test_boundary_conditions_with_data_bit_exact(model, scheme, perturb, temporal):
    """BCs that read window data from Parameter_real: pressure outlet (bvbs_outpres) and isothermal wall
    (bvbs_wall_viscous_isothermal) of the compiled reference, Cartesian and curvilinear, 5 and 6 equations."""
    from common import data_bcs
    n = (9, 8, 7)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    bcs = data_bcs(n, model)
    if temporal == 'implicit':
        bcs['jmax'] = 'BCWall'    # a face treated implicitly next to the outlets
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb,
                  temporal_scheme=temporal, time_step=0.02 if temporal == 'implicit' else None)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    for it in range(1, 5):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all()


@pytest.mark.parametrize("scheme,model,bcs,blocks", [
    ('senseur', 'NSLaminar', None, (1, 1, 1)),
    ('ausmpred', 'Euler', None, (1, 1, 1)),
    ('roe', 'NSLaminar', None, (1, 1, 1)),
    ('senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCWall'}, (1, 1, 1)),
    ('ausmpred', 'Euler', {'kmin': 'BCWallInviscid', 'kmax': 'BCWallInviscid', 'imax': 'BCSymmetryPlane'}, (1, 1, 1)),
    ('senseur', 'NSLaminar', FAR, (1, 1, 1)),
    ('ausmpred', 'NSLaminar', None, (2, 1, 2)),
])
def test_curvilinear_zone_bit_exact(scheme, model, bcs, blocks):
    """Zone type 0 (general metrics per cell): fluxes of fluausm/flusenseur/fluroe *_3dfull.for and
    fluvisq_3dfull.for, BC normals from the face metrics, on a smoothly displaced mesh."""
    nn = (10, 8, 8)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t = cases.tgv(nn, scheme, model=model, extra_numz=extra, bcs=bcs, blocks=blocks, perturb=0.15)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    so, sr = _pair(t, blocks if blocks != (1, 1, 1) else None, nv)
    assert int(so.zones[0].pint[25]) == 0
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-6


@pytest.mark.parametrize("scheme,model,perturb,bcs,blocks", [
    ('ausmpred', 'NSLaminar', 0., None, (1, 1, 1)),
    ('senseur', 'NSLaminar', 0., None, (1, 1, 1)),
    ('roe', 'Euler', 0., None, (1, 1, 1)),
    ('senseur', 'Euler', 0.15, None, (1, 1, 1)),
    ('ausmpred', 'NSLaminar', 0.15, None, (1, 1, 1)),
    ('roe', 'NSLaminar', 0.15, None, (1, 1, 1)),
    ('ausmpred', 'NSLaminar', 0., FAR, (1, 1, 1)),
    ('senseur', 'NSLaminar', 0., None, (2, 1, 2)),
    ('ausmpred', 'NSLaminar', 0.15, None, (1, 2, 1)),
    # walls: the residual is mirrored into the ghost cells and the LU range grows over them
    ('ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, (1, 1, 1)),
    ('senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, (1, 1, 1)),
    ('senseur', 'Euler', 0.15, {'jmin': 'BCWall', 'kmax': 'BCWallInviscid', 'imin': 'BCFarfield', 'imax': 'BCSymmetryPlane'}, (1, 1, 1)),
    ('ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWall'}, (2, 1, 1)),
])
def test_implicit_lusgs_bit_exact(scheme, model, perturb, bcs, blocks):
    """Implicit time step (BDF1 then BDF2 Newton residual of core3as2.for, coefficients of tstb3c.for,
    LU-SGS sweeps + Newton update of LU/invlu_l.for, invlu_u.for, mjr_newton.for) against the
    compiled reference driven through navier_stokes_struct_ and invlu_."""
    nn = (10, 8, 8)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t = cases.tgv(nn, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 4}, bcs=bcs, blocks=blocks,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    so, sr = _pair(t, blocks if blocks != (1, 1, 1) else None, nv)
    assert so.zones[0].implicit and so.zones[0].nsub() == 4
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):    # the three time levels start equal
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro) and np.array_equal(a.ro_m1, b.ro_m1), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-4


@pytest.mark.parametrize("scheme,model,perturb,bcs,blocks,nb_relax", [
    ('ausmpred', 'NSLaminar', 0., None, (1, 1, 1), 2),
    ('senseur', 'Euler', 0., None, (1, 1, 1), 3),
    ('senseur', 'NSLaminar', 0.15, None, (1, 1, 1), 2),
    ('roe', 'NSLaminar', 0.15, FAR, (1, 1, 1), 2),
    ('ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, (1, 1, 1), 2),
    ('senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, (1, 1, 1), 3),
    ('ausmpred', 'NSLaminar', 0., None, (1, 2, 1), 2),
])
def test_implicit_lussor_bit_exact(scheme, model, perturb, bcs, blocks, nb_relax):
    """LU-SSOR with nb_relax > 1 relaxations (LU/invlu.for:195-340, invlussor1_l.for, invlussor_l.for,
    invlussor_u.for, mjr_ssor.for) against invlu_ of the compiled reference."""
    nn = (10, 8, 9)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000, 'nb_relax': nb_relax}
    t = cases.tgv(nn, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs, blocks=blocks,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    so, sr = _pair(t, blocks if blocks != (1, 1, 1) else None, nv)
    assert int(so.zones[0].pint[71]) == nb_relax
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro) and np.array_equal(a.ro_m1, b.ro_m1), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-4
    # the relaxations matter: the same case with nb_relax = 1 gives another field
    extra['nb_relax'] = 1
    t1 = cases.tgv(nn, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs, blocks=blocks,
                   temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    s1, _ = _pair(t1, blocks if blocks != (1, 1, 1) else None, nv)
    s1.fill_ghostcells()
    for a in s1.zones:
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro
    for it in range(1, 5):
        s1.step()
    assert not np.array_equal(s1.zones[0].ro, so.zones[0].ro)


@pytest.mark.parametrize("nn,scheme,model,perturb,bcs,nblk", [
    ((10, 8, 8), 'ausmpred', 'NSLaminar', 0., None, 1),
    ((12, 23, 21), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'imax': 'BCFarfield'}, 4),   # 1 x 2 x 2 sub-blocks
    ((9, 24, 31), 'roe', 'NSTurbulent', 0., {'jmin': 'BCWall'}, 6),                              # 1 x 2 x 3, 6 equations
])
def test_residual_history_vs_cprdu3s1(nn, scheme, model, perturb, bcs, nblk):
    """Residual table of a verification step: the L2 / Loo norms of every Newton sub-iteration (verification slot
    included) from cprdu3s1_ of the compiled reference (per sub-block sums of Compute/cprdu3s1_incl.for:47-141,
    reduced as DISPLAY/display_ss_iteration.cpp:112-146) against the restatement: same sub-blocks, same summation
    order, equality."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t = cases.tgv(nn, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 4}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    so, sr = _pair(t)
    assert int(np.prod(O.residual_blocks(so.zones[0].pint))) == nblk     # sub-blocks of 100 x 10 x 10 cells
    sr.verif = True
    for z in so.zones:
        z.verif = True
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    for it in range(1, 4):
        so.step(); sr.step(it)
        l2r, loor = sr.residual_history()[0]
        l2o, looo = so.zones[0].residual_history()
        assert l2r.shape == (5, so.zones[0].neq)
        assert np.array_equal(l2o, l2r) and np.array_equal(looo, loor), "step %d" % it
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    assert l2r[4, 0] < l2r[0, 0]       # the Newton iterations converge


@pytest.mark.parametrize("scheme,model,perturb,bcs", [
    ('ausmpred', 'NSLaminar', 0., None),
    ('senseur', 'Euler', 0., None),
    ('roe', 'NSLaminar', 0.15, None),
    ('ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCWall'}),
])
def test_implicit_local_time_step_bit_exact(scheme, model, perturb, bcs):
    """time_step_nature='local': dt of every cell = CFL * min(convective, viscous limit) (tstb3_impli.for,
    dtloc.for), BDF1 weights and unit diagonal shift at every step (core3as2.for:37-41)."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000,
             'time_step_nature': 'local', 'cfl': 4.}
    t = cases.tgv((10, 8, 7), scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[46]) == 1
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-4


@pytest.mark.parametrize("scheme,bcs,blocks", [
    ('ausmpred', None, (1, 1, 1)),
    ('senseur', None, (1, 1, 1)),
    ('ausmpred', {'jmin': 'BCWall', 'jmax': 'BCWall'}, (1, 1, 1)),
    ('senseur', {'jmin': 'BCWall', 'kmax': 'BCFarfield', 'imin': 'BCExtrapolate', 'imax': 'BCWallInviscid'}, (1, 1, 1)),
    ('ausmpred', {'jmin': 'BCWall'}, (2, 1, 1)),
    ('roe', None, (1, 1, 1)),
])
def test_spalart_allmaras_explicit_bit_exact(scheme, bcs, blocks):
    """NSTurbulent on Cartesian zones: total viscosity (vispalart), source term (spsource_SA), convective and
    diffusive flux of rho*nu_tilde, turbulent conductivity (mut_prandtltb_interface), clipped update
    (core3ark3.for:196-236), 6-variable BCs."""
    nn = (10, 8, 8)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t = cases.tgv(nn, scheme, model='NSTurbulent', extra_numz=extra, bcs=bcs, blocks=blocks)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    so, sr = _pair(t, blocks if blocks != (1, 1, 1) else None, nv)
    assert so.zones[0].neq == 6
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro), "step %d" % it
    d = np.abs(so.zones[0].ro - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(so.zones[0].ro).all() and d[5] > 1e-7 and d[1] > 1e-4


@pytest.mark.parametrize("scheme,bcs,blocks,local", [
    ('ausmpred', None, (1, 1, 1), False),
    ('senseur', None, (1, 1, 1), False),
    ('ausmpred', {'jmin': 'BCWall', 'jmax': 'BCWall'}, (1, 1, 1), False),
    ('roe', None, (1, 1, 1), False),
    ('ausmpred', {'jmin': 'BCWall', 'kmax': 'BCFarfield', 'imin': 'BCExtrapolate', 'imax': 'BCWallInviscid'}, (1, 1, 1), True),
    ('ausmpred', {'jmin': 'BCWall'}, (2, 1, 1), False),
])
def test_spalart_allmaras_implicit_bit_exact(scheme, bcs, blocks, local):
    """NSTurbulent + implicit on Cartesian zones: 6-equation Newton residual (core3as2.for:49-231), diagonal of the
    nu_tilde equation (SA/sourceSA_LU.for), scalar 6th equation of the sweeps (LU/lu_*_3dcart_SA.for,
    lu_dinv_SA.for, lu_d_SA.for) and the clipped Newton update (LU/mjr_newton_SA.for)."""
    nn = (10, 8, 8)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if local:
        extra.update({'time_step_nature': 'local', 'cfl': 4.})
    t = cases.tgv(nn, scheme, model='NSTurbulent', extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  blocks=blocks, temporal_scheme='implicit', time_step=0.02)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    so, sr = _pair(t, blocks if blocks != (1, 1, 1) else None, nv)
    assert so.zones[0].neq == 6 and so.zones[0].implicit
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro), "step %d: %s" % (it, np.abs(a.ro - b.ro).reshape(6, -1).max(axis=1))
    d = np.abs(so.zones[0].ro - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(so.zones[0].ro).all() and d[5] > 1e-7 and d[1] > 1e-4


@pytest.mark.parametrize("scheme,model,perturb,bcs,implicit", [
    ('ausmpred', 'NSLaminar', 0.15, None, False),
    ('senseur', 'NSLaminar', 0.15, None, False),
    ('roe', 'Euler', 0.15, None, False),
    ('senseur', 'Euler', 0., None, False),
    ('ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    ('senseur', 'Euler', 0.15, {'jmin': 'BCWallInviscid', 'imax': 'BCExtrapolate', 'imin': 'BCFarfield'}, False),
    ('ausmpred', 'NSLaminar', 0.15, None, True),
    ('roe', 'NSLaminar', 0.15, None, True),
    ('senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    ('ausmpred', 'Euler', 0., {'jmin': 'BCWallInviscid', 'jmax': 'BCFarfield'}, 'local'),
])
def test_two_dimensional_zone_bit_exact(scheme, model, perturb, bcs, implicit):
    """Zone type 3 (one cell along k, ti/tj with two components): 2d flux fragments (fluFaceEuler_o3_2d_*,
    fluvisq_2d_*), update without w (core3ark3.for:156-183), sensor on i/j faces, and with the implicit scheme
    tstb3_impli.for:315-380, LU/lu_{i,j}_2d.for, mjr_newton_2d.for."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    kw = {}
    if implicit:
        kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})
        if implicit == 'local':
            extra.update({'time_step_nature': 'local', 'cfl': 4.})
    t = cases.tgv2d((10, 8), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[25]) == 3
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(5, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-5


@pytest.mark.parametrize("dim,scheme,bcs,implicit", [
    (3, 'ausmpred', None, False),
    (3, 'senseur', None, False),
    (3, 'roe', None, False),
    (3, 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    (3, 'ausmpred', None, True),
    (3, 'roe', None, True),
    (3, 'senseur', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    (2, 'ausmpred', None, False),
    (2, 'senseur', None, False),
    (2, 'roe', None, False),
    (2, 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, False),
    (2, 'ausmpred', None, True),
    (2, 'roe', None, True),
    (2, 'senseur', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
    (2, 'ausmpred', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 'local'),
])
def test_spalart_allmaras_general_mesh_bit_exact(dim, scheme, bcs, implicit):
    """NSTurbulent on curvilinear (type 0) and two-dimensional (type 3) zones: SA/sourceSA_grad_3d.for /
    sourceSA_grad_2d.for, SA/fluvisq_3dfull_*.for + fluvisq_SA_3dfull_*.for (2d: fluvisq_SA_2d_*.for),
    ROE|AUSM|SENSOR/fluFaceSA.for, and with the implicit scheme the 6-equation LU-SGS."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    kw = {}
    if implicit:
        kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})
        if implicit == 'local':
            extra.update({'time_step_nature': 'local', 'cfl': 4.})
    if dim == 3:
        t = cases.tgv((10, 8, 8), scheme, model='NSTurbulent', extra_numz=extra, bcs=bcs, perturb=0.15, **kw)
    else:
        t = cases.tgv2d((10, 8), scheme, model='NSTurbulent', extra_numz=extra, bcs=bcs, perturb=0.15, **kw)
    so, sr = _pair(t)
    assert so.zones[0].neq == 6 and int(so.zones[0].pint[25]) == (0 if dim == 3 else 3)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(6, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    d = np.abs(so.zones[0].ro - ro0).reshape(6, -1).max(axis=1)
    assert np.isfinite(so.zones[0].ro).all() and d[5] > 1e-7 and d[1] > 1e-4


@pytest.mark.parametrize("scheme,model,temporal", [
    ('ausmpred', 'NSTurbulent', 'implicit'),
    ('senseur', 'NSTurbulent', 'implicit'),
    ('ausmpred', 'NSLaminar', 'implicit'),
    ('ausmpred', 'NSTurbulent', 'explicit'),
])
def test_naca0012_cmesh_bit_exact(scheme, model, temporal):
    """Steady-RANS set-up on a two-dimensional C-mesh: Spalart-Allmaras, implicit LU-SGS with a local time step,
    wall on the section, far field outside, match join across the wake cut (the zone is its own donor)."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if temporal == 'explicit':
        extra['time_step'] = 2.e-5
    nf, nw, nj = 16, 6, 10
    t = cases.naca_cmesh(nfoil=nf, nwake=nw, nj=nj, scheme=scheme, model=model, temporal_scheme=temporal, extra_numz=extra)
    metrics = prepare(t)
    z = Internal.getZones(t)[0]
    pint, preal, xyz = zone_arrays(z)
    rp, dp = connect.cmesh_wake_join(2 * nf + 2 * nw, nj, nw)
    so = O.OracleSolver([O.OracleZone(pint, preal, FastC.level_buffer(z), metrics[0][0], wall_distance(z))], [(0, rp, 0, dp)])
    sr = R.RefSolver([R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], FastC.level_buffer(z), metrics[0][0], wall_distance(z))],
                     [(0, rp, 0, dp)])
    assert int(pint[25]) == 3
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 7):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(so.zones[0].neq, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-4


@pytest.mark.parametrize("model,perturb,dim,bcs,implicit", [
    ('Euler', 0., 3, {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, False),
    ('NSLaminar', 0., 3, {'jmin': 'BCWall', 'kmax': 'BCWallViscous', 'imin': 'BCSymmetryPlane'}, False),
    ('NSLaminar', 0.15, 3, {'jmin': 'BCWall', 'jmax': 'BCWallInviscid', 'imax': 'BCFarfield'}, False),
    ('NSTurbulent', 0.15, 3, {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, False),
    ('NSLaminar', 0.15, 2, {'jmin': 'BCWall', 'jmax': 'BCFarfield', 'imin': 'BCWallInviscid'}, False),
    ('NSLaminar', 0., 3, {'jmin': 'BCWall', 'jmax': 'BCWallInviscid'}, True),
    ('NSTurbulent', 0.15, 2, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, True),
])
def test_roe_wall_flux_correction_bit_exact(model, perturb, dim, bcs, implicit):
    """roe + walls: the flux through a wall face is replaced by the wall flux (BC/bfl3.for, bflwall.for,
    bflwallslip.for, corr_fluroe_euler_o3_*.for)."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    kw = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3}) if implicit else {}
    if dim == 3:
        t = cases.tgv((9, 8, 7), 'roe', model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv2d((10, 8), 'roe', model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(so.zones[0].neq, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-5


@pytest.mark.parametrize("blocks,nn", [((2, 1, 1), (12, 6, 6)), ((1, 1, 3), (6, 6, 15)), ((2, 2, 2), (8, 8, 8))])
def test_match_joins_bit_exact(blocks, nn):
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t = cases.tgv(nn, 'senseur', blocks=blocks, extra_numz=extra)
    so, sr = _pair(t, blocks, nv)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for it in range(1, 4):
        so.step(); sr.step(it)
        for a, b in zip(so.zones, sr.zones):
            assert np.array_equal(a.ro, b.ro), "step %d" % it


def test_split_box_equals_single_block_to_rounding():
    """An inviscid periodic box split in blocks and joined gives the single-block solution up to
    rounding.  (With viscous terms the two differ at O(1e-10): the cross derivatives next to a
    block edge read edge ghost cells, which are averages of face ghosts in a block but true
    periodic images along the interior of a single block.)"""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    t1 = cases.tgv(8, 'ausmpred', model='Euler', extra_numz=extra)
    s1, _ = _pair(t1)
    t2 = cases.tgv(8, 'ausmpred', model='Euler', blocks=(1, 1, 2), extra_numz=extra)
    s2, _ = _pair(t2, (1, 1, 2), (8, 8, 4))
    s1.fill_ghostcells(); s2.fill_ghostcells()
    for it in range(5):
        s1.step(); s2.step()
    a = s1.zones[0].ro.reshape(5, -1)[:, :12 * 12 * 12].reshape(5, 12, 12, 12)[:, 2:-2, 2:-2, 2:-2]
    lo = s2.zones[0].ro.reshape(5, -1)[:, :8 * 12 * 12].reshape(5, 8, 12, 12)[:, 2:-2, 2:-2, 2:-2]
    hi = s2.zones[1].ro.reshape(5, -1)[:, :8 * 12 * 12].reshape(5, 8, 12, 12)[:, 2:-2, 2:-2, 2:-2]
    b = np.concatenate([lo, hi], axis=1)
    assert np.abs(a - b).max() < 1e-12


IMPL = dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})
WALLS = {'jmin': 'BCWall', 'jmax': 'BCFarfield', 'kmin': 'BCWall'}


@pytest.mark.parametrize("n,scheme,model,bcs,kw,blocks", [
    ((9, 8, 7), 'ausmpred', 'NSLaminar', None, {}, None),
    ((12, 10, 9), 'senseur', 'NSLaminar', WALLS, {}, None),
    ((10, 9, 8), 'roe', 'Euler', None, {'perturb': 0.15}, None),
    ((10, 9, 8), 'senseur', 'NSLaminar', WALLS, {'perturb': 0.15}, None),
    ((10, 9, 8), 'ausmpred', 'NSLaminar', WALLS, {'perturb': 0.2, 'homo': True}, None),
    ((16, 9, 12), 'senseur', 'NSLaminar', None, {}, (2, 1, 2)),
    ((16, 9, 12), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall'}, {'perturb': 0.15}, (2, 1, 2)),
    ((10, 9, 8), 'senseur', 'NSLaminar', WALLS, {'perturb': 0.15, 'nb_relax': 2}, None),     # with LU-SSOR relaxations
])
def test_implicit_lu_match_bit_exact(n, scheme, model, bcs, kw, blocks):
    """lu_match = 1 (LU/invlu.for:98-109: invlu_overlap_l.for, invlu_overlap_u.for) with one thread per zone, against
    invlu_ of the compiled reference.  The reference also copies the Newton residual into the ghost cells of joins
    first (lhs.cpp:1-15); neither side does here, and the second half of the test shows why that is the same: the
    reference's sweeps give the same bits with those ghost cells set to 1e30."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000, 'lu_match': 1}
    kw = dict(kw)
    ssor = 'nb_relax' in kw
    if ssor:
        extra['nb_relax'] = kw.pop('nb_relax')
    nv = tuple(n[d] // blocks[d] for d in range(3)) if blocks else None

    def run(poison, nit):
        t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, blocks=blocks or (1, 1, 1), **kw, **IMPL)
        so, sr = _pair(t, blocks, nv)
        assert int(so.zones[0].pint[75]) == 1
        if poison:
            orig = sr._bczone_rhs

            def poisoned(nd, zn, nstep):
                p = zn.pint
                ni, nj, nk, neq, ndx = int(p[0]), int(p[1]), int(p[2]), int(p[36]), int(p[41])
                d = zn.drodm.reshape(neq, ndx)[:, :ni * nj * nk].reshape(neq, nk, nj, ni)
                keep = d[:, 2:-2, 2:-2, 2:-2].copy()
                d[...] = 1.e30
                d[:, 2:-2, 2:-2, 2:-2] = keep
                return orig(nd, zn, nstep)
            sr._bczone_rhs = poisoned
        so.fill_ghostcells(); sr.fill_ghostcells('ro')
        for it in range(1, nit + 1):
            so.step(); sr.step(it)
            for a, b in zip(so.zones, sr.zones):
                assert np.array_equal(a.ro, b.ro), "step %d" % it
        return so.zones[0].ro.copy()

    got = run(False, 5)
    assert np.isfinite(got).all()
    if blocks:
        run(True, 2)
    # not the sweeps of lu_match = 0: the terms are added in another order
    extra['lu_match'] = 0
    t0 = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, blocks=blocks or (1, 1, 1), **kw, **IMPL)
    s0, _ = _pair(t0, blocks, nv)
    s0.fill_ghostcells()
    for it in range(5):
        s0.step()
    if ssor:      # the relaxations of LU-SSOR are the same sweeps whatever lu_match is (LU/invlu.for:195-340)
        assert np.array_equal(s0.zones[0].ro, got)
    else:
        assert not np.array_equal(s0.zones[0].ro, got)
        np.testing.assert_allclose(s0.zones[0].ro, got, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n,scheme,model,bcs,perturb,init,verif", [
    ((9, 8, 7), 'ausmpred', 'NSLaminar', None, 0., 1, False),
    ((10, 9, 8), 'senseur', 'NSLaminar', {'jmin': 'BCWall'}, 0.15, 3, False),
    ((10, 9, 8), 'roe', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 2, False),
    ((10, 9, 8), 'senseur', 'NSLaminar', {'jmin': 'BCWall'}, 0.15, 2, True),
])
def test_selective_frequency_damping_bit_exact(n, scheme, model, bcs, perturb, init, verif):
    """sfd.for, once per implicit time step in its last slot that runs (lhs.cpp:181-185): after the LU solve of the last
    Newton sub-iteration, or, on a verification step, after the residual of the verification slot (then no join or BC
    follows).  The restatement against sfd_ of the compiled reference: state and filtered state."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000,
             'sfd': 1, 'sfd_chi': 0.5, 'sfd_delta': 0.3, 'sfd_init_iter': init}
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **IMPL)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[62]) == 1 and int(so.zones[0].pint[63]) == init
    so.zones[0].verif = verif; sr.verif = verif
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    before = None
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
        assert np.array_equal(so.zones[0].rof, sr.zones[0].rof), "filtered state, step %d" % it
    assert np.abs(so.zones[0].rof).max() > 0.1
    # the damping acts: not the undamped run
    extra['sfd'] = 0
    t0 = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **IMPL)
    s0, _ = _pair(t0)
    s0.zones[0].verif = verif
    s0.fill_ghostcells()
    for it in range(5):
        s0.step()
    assert np.abs(s0.zones[0].ro - so.zones[0].ro).max() > 1e-6


@pytest.mark.parametrize("n,scheme,model,bcs,perturb,init,rk", [
    ((12, 10, 9), 'senseur', 'NSLaminar', None, 0., 1, 3),
    ((10, 9, 8), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 3, 3),
    ((10, 9, 8), 'roe', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 2, 3),
    ((14, 9, 8), 'roe', 'Euler', None, 0., 2, 2),
])
def test_selective_frequency_damping_explicit_bit_exact(n, scheme, model, bcs, perturb, init, rk):
    """The explicit loop runs the condition of lhs.cpp:181-185 with no verification slot: sfd_ after the last stage but
    one (nitcfg = nssiter - 1), on the level that stage has written, before its joins and BCs."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000,
             'sfd': 1, 'sfd_chi': 0.5, 'sfd_delta': 0.3, 'sfd_init_iter': init}
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, time_step=0.02,
                  extra_numb={'rk': rk} if rk != 3 else None)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[62]) == 1 and int(so.zones[0].pint[58]) == rk
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
        assert np.array_equal(so.zones[0].rof, sr.zones[0].rof), "filtered state, step %d" % it
    assert np.abs(so.zones[0].rof).max() > 0.1


@pytest.mark.parametrize("dim,n,scheme,model,bcs,perturb,nb_relax,lu_match", [
    (3, (10, 9, 8), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 2, 0),
    (3, (10, 9, 8), 'ausmpred', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 2, 1),
    (3, (10, 9, 8), 'roe', 'NSTurbulent', None, 0.15, 1, 1),
    (3, (10, 9, 8), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 1, 1),
    (2, (14, 12), 'ausmpred', 'NSLaminar', None, 0.15, 2, 0),
    (2, (14, 12), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 1, 1),
    (2, (40, 11), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 2, 1),
    (2, (40, 11), 'roe', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 1, 1),
    (2, (14, 12), 'ausmpred', 'Euler', None, 0., 3, 1),
])
def test_ssor_and_lu_match_six_equations_and_two_dimensional(dim, n, scheme, model, bcs, perturb, nb_relax, lu_match):
    """LU-SSOR relaxations and lu_match = 1 with 6 equations (LU/invlussor1_l_SA.for, invlussor_l_SA.for,
    invlussor_u_SA.for, invlu_overlap_l_SA.for, invlu_overlap_u_SA.for) and on two-dimensional zones (their 2d branches),
    against invlu_ of the compiled reference."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000,
             'lu_match': lu_match, 'nb_relax': nb_relax}
    kw = dict(model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **IMPL)
    t = cases.tgv(n, scheme, **kw) if dim == 3 else cases.tgv2d(n, scheme, **kw)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all()


@pytest.mark.parametrize("n,scheme,model,bcs,opts", [
    ((9, 8, 7), 'senseur', 'NSLaminar', None, {}),
    ((12, 10, 9), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCWall'}, {}),
    ((9, 8, 7), 'roe', 'NSLaminar', None, {}),      # a case where multiplying before subtracting changes a last bit
    ((9, 8, 7), 'roe', 'NSLaminar', {'jmin': 'BCWall', 'kmax': 'BCWall'}, {}),
    ((9, 8, 7), 'roe', 'Euler', {'imin': 'BCWallInviscid', 'jmax': 'BCFarfield'}, {}),
    ((9, 8, 7), 'senseur', 'Euler', None, {}),
    ((9, 8, 7), 'senseur', 'NSTurbulent', {'jmin': 'BCWall'}, {}),
    ((10, 9, 8), 'roe', 'NSTurbulent', None, {}),
    ((9, 8, 7), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'kmin': 'BCFarfield'}, IMPL),
    ((9, 8, 7), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, IMPL),
    ((10, 9, 8), 'roe', 'NSLaminar', {'jmin': 'BCWall'}, dict(IMPL, extra_numz={'nb_relax': 2})),
])
def test_homogeneous_zone_bit_exact(n, scheme, model, bcs, opts):
    """k-homogeneous zones (zone type 1): the restatement runs the general formulas on zero-padded metrics of one
    plane, with the z derivatives of the viscous flux and of the SA source factored as fluvisq_3dhomo_{i,j,k}.for and
    SA/sourceSA_grad_3dhomo.for write them; the compiled reference runs its own *_3dhomo_* fragments (fluxes,
    tstb3_impli.for:175-255, lu_{i,j,k}_3dhomogene.for, cpcfl3.for:142-187)."""
    opts = dict(opts)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    extra.update(opts.pop('extra_numz', {}))
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=0.2, homo=True, **opts)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[25]) == 1
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all()


@pytest.mark.parametrize("n,perturb,homo", [((8, 8, 8), 0., False), ((5, 9, 7), 0., False), ((7, 9, 6), 0.2, False),
                                            ((7, 9, 6), 0.2, True), ((11, 5, 8), 0.3, True)])
def test_host_metrics_equal_reference(n, perturb, homo):
    """ti,tj,tk,vol built on the host equal the reference's cp_tijk / cp_vol bit for bit, and the
    zone classification equals nature_geom_dom (Cartesian, curvilinear and k-homogeneous zones)."""
    t = cases.tgv(n, 'senseur', perturb=perturb, homo=homo)
    metrics = prepare(t)
    z = Internal.getZones(t)[0]
    pint, preal, xyz = zone_arrays(z)
    assert np.array_equal(metrics[0][0], R.ref_metrics(pint, preal, xyz[0], xyz[1], xyz[2]))
    typ, nijk_mtr, ndimdx_mtr, neq_ij, neq_k = R.ref_nature_geom_dom(xyz[0], xyz[1], xyz[2], 2, 2)
    assert typ == int(pint[25]) == ((1 if homo else 0) if perturb else 2)
    assert nijk_mtr == [int(v) for v in pint[5:10]]
    assert ndimdx_mtr == int(pint[43]) and neq_ij == int(pint[37]) and neq_k == int(pint[38])


@pytest.mark.parametrize("kind,model,perturb", [
    ('tgv', 'NSLaminar', 0.), ('tgv', 'Euler', 0.), ('tgv', 'NSLaminar', 0.15), ('tgv', 'Euler', 0.15),
    ('tgv', 'NSTurbulent', 0.15), ('tgv2d', 'NSLaminar', 0.15), ('tgv2d', 'Euler', 0.),
])
def test_cfl_diagnostic_vs_reference(kind, model, perturb):
    """[max, min, sum] of the local CFL number against cpcfl3_ of the compiled reference, on the three zone types,
    Euler and viscous.  The reference loop carries an OpenMP SIMD directive, so its sum may be folded in a
    different order: max and min must be equal, the mean to 1e-13."""
    import ctypes
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if kind == 'tgv':
        t = cases.tgv((9, 8, 7), 'ausmpred', model=model, extra_numz=extra, perturb=perturb)
    else:
        t = cases.tgv2d((11, 9), 'ausmpred', model=model, extra_numz=extra, perturb=perturb)
    so, sr = _pair(t)
    oz, rz = so.zones[0], sr.zones[0]
    so.fill_ghostcells()
    got = oz.cfl()
    L = R.lib()
    p = oz.pint
    ind = np.array([1, p[20], 1, p[21], 1, p[22]], dtype=np.int32)
    ref = np.array([0., 100000000., 0.])
    venti = np.zeros(8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L.cpcfl3_(ctypes.byref(ctypes.c_int32(0)), vp(p), vp(oz.preal), vp(ind), vp(ref), vp(oz.ro), vp(oz.xmut), vp(venti),
              vp(rz.ti), vp(rz.tj), vp(rz.tk), vp(rz.vol))
    ref[2] /= int(p[20]) * int(p[21]) * int(p[22])
    assert got[0] == ref[0] and got[1] == ref[1] and got[0] > got[1] > 0.
    np.testing.assert_allclose(got[2], ref[2], rtol=1e-13)


@pytest.mark.parametrize("kind,n,perturb,collapse,model", [
    ('tgv', (9, 8, 7), 0.15, ('jmin',), 'NSLaminar'),
    ('tgv', (8, 7, 9), 0.15, ('jmin', 'jmax', 'kmax', 'imin'), 'NSTurbulent'),
    ('tgv', (7, 9, 8), 0.2, ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax'), 'NSLaminar'),
    ('tgv2d', (11, 9), 0.15, ('jmin', 'imax'), 'NSTurbulent'),
    ('tgv', (8, 8, 8), 0., ('jmin', 'kmin'), 'NSLaminar'),     # stays Cartesian: collapsed cells are not tested
])
def test_collapsed_ghost_cells_vs_reference(kind, n, perturb, collapse, model):
    """Flat ghost cells behind physical boundaries: flags of nature_geom_dom_, metrics after tijk_extrap_ and wall
    distance after dist_extrap_ of the compiled reference against the host model (bit for bit)."""
    bcs = {f: 'BCWall' for f in collapse}
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if kind == 'tgv':
        t = cases.tgv(n, 'ausmpred', model=model, extra_numz=extra, bcs=bcs, perturb=perturb, collapse=collapse)
    else:
        t = cases.tgv2d(n, 'ausmpred', model=model, extra_numz=extra, bcs=bcs, perturb=perturb, collapse=collapse)
    z = Internal.getZones(t)[0]
    dist0 = None
    if model == 'NSTurbulent':
        dist0 = Internal.getNodeFromName1(Internal.getNodeFromName1(z, 'FlowSolution#Centers'), 'TurbulentDistance')[1].copy()
    metrics = prepare(t)
    pint, preal, xyz = zone_arrays(z)
    degen = R.ref_degenerate(xyz[0], xyz[1], xyz[2], int(pint[3]), int(pint[4]))
    assert degen.sum() > 0 and np.array_equal(degen, metrics[0][7])
    assert int(pint[25]) == (2 if perturb == 0. else (0 if kind == 'tgv' else 3))
    want = R.ref_tijk_extrap(pint, degen, R.ref_metrics(pint, preal, xyz[0], xyz[1], xyz[2]))
    assert np.array_equal(want, metrics[0][0])
    if int(pint[25]) != 2:
        assert not np.array_equal(want, R.ref_metrics(pint, preal, xyz[0], xyz[1], xyz[2]))
    if dist0 is not None:
        got = Internal.getNodeFromName1(Internal.getNodeFromName1(z, 'FlowSolution#Centers'), 'TurbulentDistance')[1]
        assert np.array_equal(R.ref_dist_extrap(pint, degen, dist0), got.reshape(-1, order='F'))
    # and the flow solver runs on them
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for it in range(1, 4):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all()


@pytest.mark.parametrize("kind,scheme,model,perturb,bcs", [
    ('tgv', 'ausmpred', 'NSLaminar', 0., None),
    ('tgv', 'senseur', 'Euler', 0.15, None),
    ('tgv', 'roe', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
    ('tgv', 'ausmpred', 'NSTurbulent', 0., {'jmin': 'BCWall'}),
    ('tgv2d', 'ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
])
def test_explicit_local_time_step_bit_exact(kind, scheme, model, perturb, bcs):
    """time_step_nature='local' with the explicit scheme: every cell advances with its own CFL-limited step
    (cptst3.for:77-82 -> tstb3_expli.for, dtloc.for) in the Runge-Kutta update."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000,
             'time_step_nature': 'local', 'cfl': 0.5}
    if kind == 'tgv':
        t = cases.tgv((9, 8, 7), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb)
    else:
        t = cases.tgv2d((11, 9), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[46]) == 1 and int(so.zones[0].pint[29]) == 2
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 6):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d" % it
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-6


@pytest.mark.parametrize("n,scheme,threads", [(40, 'senseur', 4), ((20, 40, 44), 'ausmpred', 8)])
def test_reference_on_threads_agrees_with_one_thread(n, scheme, threads):
    """The CPU timing baseline runs the compiled reference on several OpenMP threads (oracle/ref_mt.c: its own thread
    sub-domains and lock protocol).  The flux scatter then reaches the cells next to a sub-domain boundary in another
    order, so the fields agree with the one-thread run to rounding, not bit for bit (on a coarse mesh a last-bit
    change can flip the 0/1 sensor of the 'senseur' scheme and grow: the box here is resolved enough not to)."""
    t = cases.tgv(n, scheme)
    _, ref1 = _pair(t)
    t = cases.tgv(n, scheme)
    _, refn = _pair(t)
    refn.threads = threads
    for s in (ref1, refn):
        s.fill_ghostcells('ro')
        for it in range(1, 4):
            s.step(it)
    a, b = ref1.zones[0].ro, refn.zones[0].ro
    assert np.isfinite(b).all()
    assert np.abs(a - b).max() < 1e-13 * np.abs(a).max()


@pytest.mark.parametrize("scheme,model,perturb,dim,bcs,kw", [
    ('roe_min', 'NSLaminar', 0., 3, None, {}),
    ('roe_nul', 'NSLaminar', 0., 3, None, {}),
    ('roe_min', 'Euler', 0., 3, {'jmin': 'BCWall', 'jmax': 'BCWallInviscid', 'imin': 'BCFarfield'}, {}),
    ('roe_min', 'NSLaminar', 0.15, 3, {'jmin': 'BCWall', 'kmax': 'BCWallViscous'}, {}),
    ('roe_nul', 'Euler', 0.15, 3, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {}),
    ('roe_min', 'NSTurbulent', 0.15, 3, {'jmin': 'BCWall'}, {}),
    ('roe_min', 'NSLaminar', 0.15, 2, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {}),
    ('roe_nul', 'NSLaminar', 0.15, 2, None, {}),
    ('roe_min', 'NSLaminar', 0.2, 3, {'jmin': 'BCWall'}, {'homo': True}),
    ('roe_min', 'NSLaminar', 0.15, 3, {'jmin': 'BCWall'},
     dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})),
])
def test_roe_slopes_bit_exact(scheme, model, perturb, dim, bcs, kw):
    """roe with the minmod limiter ('roe_min', slope 3) and with first-order states ('roe_nul', slope 1):
    ROE/fluroe_select.for options 1xx and 3xx, Slope/minmod_slope_var.for, Slope/o1_slope_var.for; the wall-face
    correction takes the flux of the same slope back (ROE/correction_fluroe_select.for)."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    if dim == 3:
        t = cases.tgv((9, 8, 7), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv2d((10, 8), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    so, sr = _pair(t)
    assert so.zones[0].pint[33] == 5 and so.zones[0].pint[64] == (3 if scheme == 'roe_min' else 1)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(so.zones[0].neq, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-5


@pytest.mark.parametrize("model,scheme,perturb,dim,bcs,kw", [
    ('NSLaminar', 'senseur', 0., 3, {'imax': 'BCOutflow', 'imin': 'BCInflow', 'jmin': 'BCWall'}, {}),
    ('Euler', 'roe', 0., 3, {'imin': 'BCOutflow', 'kmax': 'BCOutflow', 'jmax': 'BCOutflow'}, {}),
    ('NSLaminar', 'ausmpred', 0.15, 3, {'imax': 'BCOutflow', 'jmin': 'BCOutflow', 'kmin': 'BCOutflow', 'imin': 'BCFarfield'}, {}),
    ('NSLaminar', 'ausmpred', 0.15, 2, {'imax': 'BCOutflow', 'jmax': 'BCOutflow', 'jmin': 'BCWall'}, {}),
    ('NSLaminar', 'senseur', 0.15, 3, {'imax': 'own', 'jmin': 'BCWall'}, {}),
    ('NSLaminar', 'ausmpred', 0.15, 3, {'imax': 'BCOutflow', 'kmin': 'BCOutflow', 'jmin': 'BCWall'},
     dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})),
])
def test_outflow_bit_exact(model, scheme, perturb, dim, bcs, kw):
    """BCOutflow (tag 10, bvbs_outflow.for, BCOutflow.for): the reference pressure with the rest taken from inside, through
    the characteristic relations of non_reflection.for; on the reference state or on the BC's own uniform state."""
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    cv = 1. / (1.4 * 0.4 * 1.4 * 0.01)
    if bcs.get('imax') == 'own':
        ro, u, T = 0.97, 0.8, 1.03
        st = np.array([ro, ro * u, 0., 0., ro * (cv * T + 0.5 * u * u), 0.])
        bcs = dict(bcs, imax=('BCOutflow', {'State': st}))
    if dim == 3:
        t = cases.tgv((9, 8, 7), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv2d((10, 8), scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(so.zones[0].neq, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-5


@pytest.mark.parametrize("model,scheme,perturb,temporal", [
    ('NSLaminar', 'ausmpred', 0., 'explicit'),
    ('Euler', 'roe', 0., 'explicit'),
    ('NSLaminar', 'senseur', 0.15, 'explicit'),
    ('NSTurbulent', 'ausmpred', 0.15, 'explicit'),
    ('NSLaminar', 'ausmpred', 0.15, 'implicit'),
])
def test_mass_flow_rate_bcs_bit_exact(model, scheme, perturb, temporal):
    """BCOutMFR (tag 21, bvbs_outmfr.for) and BCInjMFR (tag 20, bvbs_injmfr.for: both the quadratic for a moving state and
    the Newton start for a state at rest) of the compiled reference, Cartesian and curvilinear, 5 and 6 equations."""
    from common import mfr_bcs
    n = (9, 8, 7)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
    bcs = mfr_bcs(n, model)
    if model == 'Euler':
        bcs['jmin'] = 'BCWallInviscid'
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb,
                  temporal_scheme=temporal, time_step=0.02 if temporal == 'implicit' else None)
    so, sr = _pair(t)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    for a, b in zip(so.zones, sr.zones):
        a.ro_m1[:] = a.ro; a.ro_p1[:] = a.ro; b.ro_m1[:] = b.ro; b.ro_p1[:] = b.ro
    ro0 = so.zones[0].ro.copy()
    for it in range(1, 5):
        so.step(); sr.step(it)
        d = np.abs(so.zones[0].ro - sr.zones[0].ro).reshape(so.zones[0].neq, -1).max(axis=1)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %s" % (it, d)
    assert np.isfinite(so.zones[0].ro).all() and np.abs(so.zones[0].ro - ro0).max() > 1e-5


@pytest.mark.parametrize("dim,n,scheme,model,bcs,perturb,kw", [
    (3, (10, 8, 7), 'ausmpred', 'NSLaminar', None, 0., {}),
    (3, (10, 8, 7), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., {}),
    (3, (9, 8, 7), 'roe', 'Euler', None, 0., {}),
    (3, (10, 8, 7), 'senseur', 'Euler', None, 0.15, {}),
    (3, (10, 8, 7), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'kmax': 'BCFarfield'}, 0.15, {}),
    (3, (9, 8, 7), 'roe', 'NSLaminar', None, 0.15, {}),
    (3, (9, 8, 7), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall'}, 0.2, {'homo': True}),
    (3, (9, 8, 7), 'senseur', 'NSLaminar', None, 0.2, {'homo': True}),
    (2, (12, 10), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, {}),
    (2, (12, 10), 'senseur', 'Euler', None, 0.15, {}),
    (3, (10, 8, 7), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall'}, 0.15, IMPL),
    (3, (10, 8, 7), 'senseur', 'NSLaminar', None, 0., IMPL),
])
def test_o3sc_slope_bit_exact(dim, n, scheme, model, bcs, perturb, kw):
    """slope = 'o3sc': the o3 states weighted by the directional shock sensor of SENSOR/sciacovelli.for, evaluated at
    every sub-iteration (src_term.for:246-255), for the three schemes on the four zone types; the compiled reference
    runs its flu*_o3sc_* variants.  The velocity field is compressed along x first, so that the sensor departs from 1."""
    kw = dict(kw)
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000, 'slope': 'o3sc'}
    if dim == 2:
        t = cases.tgv2d(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    else:
        t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=perturb, **kw)
    so, sr = _pair(t)
    assert int(so.zones[0].pint[64]) == 6
    for s in (so, sr):      # a compression front: u falls by a third of the sound speed over a few cells
        zn = s.zones[0]
        nd = int(zn.pint[41]); ni = int(zn.pint[0])
        ii = np.arange(nd) % ni
        zn.ro[nd:2 * nd] -= 2. * np.tanh((ii - ni / 2.) / 1.5)
    so.fill_ghostcells(); sr.fill_ghostcells('ro')
    assert np.array_equal(so.zones[0].ro, sr.zones[0].ro)
    nd = int(so.zones[0].pint[41])
    for it in range(1, 5):
        so.step(); sr.step(it)
        assert np.array_equal(so.zones[0].ro, sr.zones[0].ro), "step %d: %g" % (it, np.abs(so.zones[0].ro - sr.zones[0].ro).max())
        w_o = so.zones[0].wig[-3 * nd:]; w_r = sr.zones[0].wig[-3 * nd:]
        assert np.array_equal(w_o, w_r), "sensor, step %d" % it
    assert np.isfinite(so.zones[0].ro).all()
    assert so.zones[0].wig[-3 * nd:].min() < 0.999, "the sensor never departed from 1"
