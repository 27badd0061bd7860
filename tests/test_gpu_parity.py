"""GPU parity: HIP path (through the C ABI) against the CPU restatement on the same inputs."""
import numpy as np
import pytest

from fast_amd import FastC, Internal, _capi
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import make_case, zone_arrays, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    ('tgv', 16, 'senseur', 'NSLaminar', 12),
    ('tgv', 16, 'ausmpred', 'NSLaminar', 12),
    ('tgv', 12, 'senseur', 'Euler', 6),
    ('tgv', 12, 'ausmpred', 'Euler', 6),
    ('tgv', 12, 'roe', 'NSLaminar', 6),
    ('vortex', 16, 'roe', None, 12),
    ('tgv', (72, 10, 37), 'senseur', 'NSLaminar', 4),     # several i-tiles, ragged j and k chunks
    ('tgv', (130, 9, 6), 'ausmpred', 'NSLaminar', 3),
    ('tgv', (64, 8, 33), 'senseur', 'NSLaminar', 3),      # tile edge passes on the zone boundary
]


@pytest.mark.parametrize("kind,n,scheme,model,nit", CASES)
def test_bit_exact_vs_oracle(kind, n, scheme, model, nit):
    t = make_case(kind, n, scheme, model)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    ro0 = FastC.level_buffer(z).copy()
    oz = O.OracleZone(pint, preal, ro0, metrics[0][0])
    # warmup already filled the ghost cells of the initial field on the device
    assert np.array_equal(ro0, oz.ro)
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), \
            "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    l2, loo = FastS.display_temporal_criteria(t, metrics, 1, format='none')[z[0]]


FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}
BC_CASES = [
    ('NSLaminar', 'senseur', (70, 9, 12), {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}),
    ('NSLaminar', 'ausmpred', (12, 10, 9), {'imin': 'BCWall', 'imax': 'BCWall', 'kmin': 'BCWallViscous'}),
    ('Euler', 'senseur', (10, 12, 9), {'jmin': 'BCWall', 'kmax': 'BCWallInviscid', 'imin': 'BCSymmetryPlane'}),
    ('NSLaminar', 'senseur', (9, 8, 10), FAR),
    ('Euler', 'roe', (8, 9, 7), {'imin': 'BCExtrapolate', 'imax': 'BCFarfield', 'kmin': 'BCExtrapolate'}),
    ('NSLaminar', 'ausmpred', (9, 8, 10), {'imin': 'BCInflow', 'imax': 'BCFarfield', 'jmin': 'BCWall', 'kmax': 'BCInflowSupersonic'}),
]


BC_CASES += [('NSLaminar', 'senseur', (9, 8, 10), 'own_state'), ('Euler', 'roe', (10, 9, 8), 'own_state')]


@pytest.mark.parametrize("model,scheme,n,bcs", BC_CASES)
def test_boundary_conditions_bit_exact(model, scheme, n, bcs):
    from fast_amd import cases
    from common import SINGLE, own_state_bcs
    if bcs == 'own_state':     # far field / inflow / supersonic inflow with their own uniform state
        bcs = own_state_bcs()
    t = cases.tgv(n, scheme, model=model, extra_numz=SINGLE, bcs=bcs)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()                                     # idempotent on a correctly filled field
    assert np.array_equal(FastC.level_buffer(z), oz.ro)      # ghost fill of the initial field
    for it in range(1, 7):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        same = np.array_equal(got, oz.ro)
        assert same, "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))


def test_unsupported_options_are_refused():
    from fast_amd import cases
    t = cases.tgv(8, 'senseur', bcs={'jmin': 'BCWallModel'})
    with pytest.raises(_capi.FastsHipError, match="tag 30"):
        FastS.warmup(t, None)
    t = cases.tgv(8, 'senseur', temporal_scheme='implicit', extra_numz={'implicit_solver': 'gmres'})
    with pytest.raises(_capi.FastsHipError, match="GMRES"):
        FastS.warmup(t, None)


@pytest.mark.parametrize("kind,model,perturb,temporal", [
    ('tgv', 'NSLaminar', 0., 'explicit'), ('tgv', 'Euler', 0., 'explicit'), ('tgv', 'NSLaminar', 0.15, 'explicit'),
    ('tgv', 'NSTurbulent', 0.15, 'implicit'), ('tgv2d', 'NSLaminar', 0.15, 'explicit'), ('tgv', 'Euler', 0.15, 'implicit'),
])
def test_cfl_diagnostic(kind, model, perturb, temporal):
    """CFL_minmaxmoy of a verification step (cpcfl3.for) against the CPU restatement (itself checked against the
    compiled reference): max and min bit for bit (1e-12 with Spalart-Allmaras: the device pow of the eddy
    viscosity), the mean to 1e-13 (the device folds the sum in another order)."""
    from fast_amd import cases
    from common import SINGLE, wall_distance
    kw = dict(model=model, extra_numz=SINGLE, perturb=perturb, temporal_scheme=temporal)
    if temporal == 'implicit':
        kw['time_step'] = 0.02
    t = cases.tgv((34, 9, 8), 'ausmpred', **kw) if kind == 'tgv' else cases.tgv2d((37, 9), 'ausmpred', **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0],
                      wall_distance(z) if int(pint[36]) == 6 else None)
    want = oz.cfl()
    FastS._compute(t, metrics, 1, None)
    FastS.display_temporal_criteria(t, metrics, 1, format='none')
    got = Internal.getNodeFromName1(Internal.getNodeFromName1(z, '.Solver#ownData'), 'CFL_minmaxmoy')[1]
    if int(pint[36]) == 6:
        np.testing.assert_allclose(got[:2], want[:2], rtol=1e-12)
    else:
        assert got[0] == want[0] and got[1] == want[1]
    np.testing.assert_allclose(got[2], want[2], rtol=1e-13)
    assert want[0] > want[1] > 0.


@pytest.mark.parametrize("kind,n,model,collapse,temporal", [
    ('tgv', (34, 9, 8), 'NSLaminar', ('jmin', 'kmax'), 'explicit'),
    ('tgv', (10, 9, 8), 'NSLaminar', ('jmin', 'jmax', 'imin'), 'implicit'),
    ('tgv2d', (37, 9), 'NSLaminar', ('jmin',), 'implicit'),
])
def test_collapsed_ghost_cells(kind, n, model, collapse, temporal):
    """Walls with flat ghost cells behind them (metrics of the ghost cells from tijk_extrap): bit for bit."""
    from fast_amd import cases
    from common import SINGLE
    bcs = {f: 'BCWall' for f in collapse}
    kw = dict(model=model, extra_numz=SINGLE, bcs=bcs, perturb=0.15, collapse=collapse, temporal_scheme=temporal)
    if temporal == 'implicit':
        kw['time_step'] = 0.02
    t = cases.tgv(n, 'senseur', **kw) if kind == 'tgv' else cases.tgv2d(n, 'senseur', **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert metrics[0][7].sum() > 0
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    if temporal == 'implicit':
        oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all()


@pytest.mark.parametrize("kind,n,scheme,model,perturb,bcs", [
    ('tgv', (34, 9, 8), 'ausmpred', 'NSLaminar', 0., None),
    ('tgv', (70, 9, 12), 'senseur', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
    ('tgv', (10, 9, 8), 'senseur', 'Euler', 0.15, None),
    ('tgv', (10, 9, 8), 'roe', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
    ('tgv', (12, 9, 8), 'roe', 'NSLaminar', 0., {'kmin': 'BCWall'}),
    ('tgv', (10, 9, 8), 'ausmpred', 'NSTurbulent', 0., {'jmin': 'BCWall'}),
    ('tgv2d', (37, 9), 'ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
])
def test_explicit_local_time_step(kind, n, scheme, model, perturb, bcs):
    """time_step_nature='local' with the explicit scheme (tstb3_expli.for): every cell advances with its own
    CFL-limited step.  Bit for bit (1e-12 with Spalart-Allmaras); the residual norms use the same per-cell step."""
    from fast_amd import cases
    from common import SINGLE, wall_distance
    extra = dict(SINGLE, time_step_nature='local', cfl=0.5)
    kw = dict(model=model, extra_numz=extra, bcs=bcs, perturb=perturb)
    t = cases.tgv(n, scheme, **kw) if kind == 'tgv' else cases.tgv2d(n, scheme, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[46]) == 1 and int(pint[29]) == 2
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if neq == 6:
            assert rel_err(got, oz.ro, int(pint[41]), neq=6).max() < 1e-12, "step %d" % it
        else:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
        if it == 1 and neq == 5:
            l2, loo = FastS.display_temporal_criteria(t, metrics, it, format='none')[z[0]]
            rl2, rloo = oz.residuals()
            np.testing.assert_allclose(l2, rl2, rtol=1e-11)
            np.testing.assert_allclose(loo, rloo, rtol=1e-13)
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-6


@pytest.mark.parametrize("scheme,model,kind,n,perturb,bcs,kw", [
    ('roe_min', 'NSLaminar', 'tgv', (70, 9, 12), 0., None, {}),                               # Cartesian zone
    ('roe_nul', 'Euler', 'tgv', (16, 9, 8), 0., {'jmin': 'BCWall', 'jmax': 'BCWallInviscid', 'imin': 'BCFarfield'}, {}),
    ('roe_min', 'Euler', 'tgv', (12, 9, 8), 0., {'jmin': 'BCWall', 'kmax': 'BCFarfield'}, {}),
    ('roe_min', 'NSLaminar', 'tgv', (10, 9, 8), 0.15, {'jmin': 'BCWall', 'kmax': 'BCWallViscous'}, {}),
    ('roe_nul', 'NSLaminar', 'tgv', (10, 9, 8), 0.15, None, {}),
    ('roe_min', 'NSTurbulent', 'tgv', (10, 9, 8), 0.15, {'jmin': 'BCWall'}, {}),
    ('roe_min', 'NSLaminar', 'tgv2d', (37, 9), 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {}),
    ('roe_min', 'NSLaminar', 'tgv', (12, 10, 9), 0.2, {'jmin': 'BCWall'}, {'homo': True}),
    ('roe_min', 'NSLaminar', 'tgv', (10, 9, 8), 0.15, {'jmin': 'BCWall'},
     dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})),
    ('roe_nul', 'NSLaminar', 'tgv', (34, 9, 8), 0., None,
     dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})),
])
def test_roe_slopes(scheme, model, kind, n, perturb, bcs, kw):
    """roe with the minmod limiter ('roe_min') and with first-order states ('roe_nul'): Slope/minmod_slope_var.for,
    Slope/o1_slope_var.for through ROE/fluroe_select.for options 3xx / 1xx; the wall-face correction takes the flux of the
    same slope back.  Bit for bit (1e-12 with Spalart-Allmaras)."""
    from fast_amd import cases
    from common import SINGLE, wall_distance
    mk = cases.tgv if kind == 'tgv' else cases.tgv2d
    t = mk(n, scheme, model=model, extra_numz=SINGLE, bcs=bcs, perturb=perturb, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[33]) == 5 and int(pint[64]) == (3 if scheme == 'roe_min' else 1)
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if neq == 6:
            assert rel_err(got, oz.ro, int(pint[41]), neq=6).max() < 1e-12, "step %d" % it
        else:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-5


def test_slopes_without_a_reference_flux_are_refused():
    from fast_amd import cases
    t = cases.tgv(8, 'ausmpred', extra_numz={'slope': 'minmod'})
    with pytest.raises(_capi.FastsHipError, match="slope"):
        FastS.warmup(t, None)
    t = cases.tgv(8, 'senseur', extra_numz={'slope': 'o1'})
    with pytest.raises(_capi.FastsHipError, match="slope"):
        FastS.warmup(t, None)


@pytest.mark.parametrize("model,scheme,kind,n,perturb,bcs,kw", [
    ('NSLaminar', 'senseur', 'tgv', (70, 9, 12), 0., {'imax': 'BCOutflow', 'imin': 'BCInflow', 'jmin': 'BCWall'}, {}),
    ('Euler', 'roe', 'tgv', (10, 9, 8), 0., {'imin': 'BCOutflow', 'kmax': 'BCOutflow', 'jmax': 'BCOutflow'}, {}),
    ('NSLaminar', 'ausmpred', 'tgv', (10, 9, 8), 0.15, {'imax': 'BCOutflow', 'jmin': 'BCOutflow', 'kmin': 'BCOutflow', 'imin': 'BCFarfield'}, {}),
    ('NSLaminar', 'ausmpred', 'tgv2d', (37, 9), 0.15, {'imax': 'BCOutflow', 'jmax': 'BCOutflow', 'jmin': 'BCWall'}, {}),
    ('NSLaminar', 'senseur', 'tgv', (10, 9, 8), 0.15, {'imax': 'own', 'jmin': 'BCWall'}, {}),
    ('NSLaminar', 'ausmpred', 'tgv', (10, 9, 8), 0.15, {'imax': 'BCOutflow', 'kmin': 'BCOutflow', 'jmin': 'BCWall'},
     dict(temporal_scheme='implicit', time_step=0.02, extra_numb={'ss_iteration': 3})),
])
def test_outflow(model, scheme, kind, n, perturb, bcs, kw):
    """BCOutflow (tag 10, bvbs_outflow.for): on the reference state and on the BC's own uniform state; bit for bit."""
    from fast_amd import cases
    from common import SINGLE
    cv = 1. / (1.4 * 0.4 * 1.4 * 0.01)
    if bcs.get('imax') == 'own':
        ro, u, T = 0.97, 0.8, 1.03
        bcs = dict(bcs, imax=('BCOutflow', {'State': np.array([ro, ro * u, 0., 0., ro * (cv * T + 0.5 * u * u), 0.])}))
    mk = cases.tgv if kind == 'tgv' else cases.tgv2d
    t = mk(n, scheme, model=model, extra_numz=SINGLE, bcs=bcs, perturb=perturb, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)      # ghost fill of the initial field
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-5


def test_outflow_with_six_equations_is_refused():
    from fast_amd import cases
    t = cases.tgv(8, 'senseur', model='NSTurbulent', bcs={'jmin': 'BCWall', 'jmax': 'BCOutflow'})
    with pytest.raises(_capi.FastsHipError, match="tag 10"):
        FastS.warmup(t, None)
