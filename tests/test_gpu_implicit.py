"""GPU parity of the implicit time step (BDF Newton residual + LU-SGS + Newton update) against the CPU
restatement, itself bit-exact against navier_stokes_struct_ / invlu_ of the compiled reference
(test_oracle_vs_ref.py).  The device sweeps hyperplanes where the reference sweeps lexicographically;
per cell the operations and their order are the same, so fields must agree bit for bit."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare

pytestmark = pytest.mark.gpu

FAR = {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}

CASES = [
    # n, scheme, model, perturb, bcs, ss_iteration, steps, layer
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0., None, 4, 4, 'c'),
    ((12, 10, 9), 'senseur', 'NSLaminar', 0., None, 3, 4, 'Python'),
    ((10, 9, 12), 'roe', 'Euler', 0., None, 3, 3, 'c'),
    ((9, 12, 10), 'senseur', 'Euler', 0., None, 2, 3, 'c'),
    ((40, 11, 35), 'ausmpred', 'NSLaminar', 0., None, 2, 2, 'c'),       # several tiles and k chunks
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0.15, None, 4, 4, 'c'),      # curvilinear
    ((10, 9, 8), 'senseur', 'NSLaminar', 0.15, None, 3, 3, 'Python'),
    ((10, 8, 9), 'roe', 'NSLaminar', 0.15, None, 2, 3, 'c'),
    ((34, 9, 36), 'senseur', 'Euler', 0.15, None, 2, 2, 'c'),
    ((9, 8, 10), 'ausmpred', 'NSLaminar', 0., FAR, 3, 4, 'c'),
    ((9, 8, 10), 'senseur', 'NSLaminar', 0.15, {'imin': 'BCExtrapolate', 'imax': 'BCFarfield'}, 3, 3, 'c'),
    # walls: BCs on the residual + LU range over the wall ghost layer
    ((12, 9, 10), 'ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 3, 4, 'c'),
    ((10, 9, 8), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 3, 3, 'Python'),
    ((9, 10, 8), 'senseur', 'Euler', 0.15, {'jmin': 'BCWall', 'kmax': 'BCWallInviscid', 'imin': 'BCFarfield',
                                            'imax': 'BCSymmetryPlane'}, 2, 3, 'c'),
    ((20, 9, 19), 'ausmpred', 'NSLaminar', 0., {'imin': 'BCWall', 'imax': 'BCWall', 'kmax': 'BCWall'}, 2, 3, 'c'),
]


@pytest.mark.parametrize("n,scheme,model,perturb,bcs,ssit,nit,layer", CASES)
def test_implicit_bit_exact_vs_oracle(n, scheme, model, perturb, bcs, ssit, nit, layer):
    t = cases.tgv(n, scheme, model=model, extra_numz=SINGLE, extra_numb={'ss_iteration': ssit}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb, modulo_verif=1)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[29]) == 1 and int(pint[58]) == ssit + 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    ro0 = oz.ro.copy()
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None, layer=layer)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
        if it == 1:   # Newton residual of the first sub-iteration
            l2, loo = FastS.display_temporal_criteria(t, metrics, it, format='none')[z[0]]
            rl2, rloo = oz.residuals()
            np.testing.assert_allclose(l2, rl2, rtol=1e-11)
            np.testing.assert_allclose(loo, rloo, rtol=1e-13)
    FastS.sync_to_host(t)
    assert np.array_equal(FastC.level_buffer(z, '_M1'), oz.ro_m1)
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4


@pytest.mark.parametrize("n,scheme,model,perturb,bcs", [
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0., None),
    ((36, 9, 10), 'senseur', 'Euler', 0., None),
    ((10, 9, 8), 'roe', 'NSLaminar', 0.15, None),
    ((10, 9, 8), 'ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCWall'}),
])
def test_implicit_local_time_step(n, scheme, model, perturb, bcs):
    """time_step_nature='local': per-cell dt from the CFL number, BDF1 at every step."""
    extra = dict(SINGLE, time_step_nature='local', cfl=4.)
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[46]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))


@pytest.mark.parametrize("blocks,nn,scheme,perturb", [
    ((2, 1, 2), (16, 9, 12), 'senseur', 0.),
    ((1, 2, 1), (10, 16, 8), 'ausmpred', 0.15),
])
def test_implicit_blocks_with_joins(blocks, nn, scheme, perturb):
    """Several zones: each runs its own LU-SGS sweep (block Jacobi across the joins, lu_match = 0), the
    Newton iterates are exchanged through the joins after every sub-iteration."""
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    kw = dict(blocks=blocks, extra_numz=SINGLE, extra_numb={'ss_iteration': 3}, temporal_scheme='implicit',
              time_step=0.02, perturb=perturb)
    to = cases.tgv(nn, scheme, **kw)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in joins])
    s.fill_ghostcells()
    t = cases.tgv(nn, scheme, **kw)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for it in range(1, 4):
        FastS._compute(t, metrics, it, tc)
        s.step()
        for z, oz in zip(zones, s.zones):
            assert np.array_equal(FastC.level_buffer(z), oz.ro), "step %d zone %s" % (it, z[0])


@pytest.mark.parametrize("n,scheme,model,perturb,bcs,nb_relax", [
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0., None, 2),
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0., None, -2),                  # with lu_match = 1: the same sweeps
    ((40, 11, 35), 'senseur', 'Euler', 0., None, 3),                   # several 8-cell units per line
    ((12, 10, 9), 'senseur', 'NSLaminar', 0.15, None, 2),              # curvilinear
    ((10, 8, 9), 'roe', 'NSLaminar', 0.15, FAR, 2),
    ((12, 9, 10), 'ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 2),
    ((10, 9, 8), 'senseur', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, 3),
])
def test_ssor_relaxations_bit_exact_vs_oracle(n, scheme, model, perturb, bcs, nb_relax):
    """LU-SSOR with nb_relax > 1 (LU/invlu.for:195-340): the restatement is bit-exact against invlu_ of the compiled
    reference (test_oracle_vs_ref.py::test_implicit_lussor_bit_exact); the device sweeps hyperplanes and resets the
    increments after a sweep instead of one plane behind it."""
    extra = dict(SINGLE, nb_relax=abs(nb_relax), lu_match=int(nb_relax < 0))
    nb_relax = abs(nb_relax)
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[71]) == nb_relax
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    ro0 = oz.ro.copy()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4


@pytest.mark.parametrize("n,scheme,model,perturb,bcs,homo", [
    ((12, 10, 9), 'ausmpred', 'NSLaminar', 0., None, False),
    ((40, 11, 35), 'senseur', 'Euler', 0., None, False),                   # several 8-cell units per line
    ((12, 10, 9), 'senseur', 'NSLaminar', 0.15, None, False),              # curvilinear
    ((10, 8, 9), 'roe', 'NSLaminar', 0.15, FAR, False),
    ((12, 9, 10), 'ausmpred', 'NSLaminar', 0., {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, False),
    ((10, 9, 8), 'senseur', 'NSLaminar', 0.2, {'jmin': 'BCWall', 'kmin': 'BCWall', 'imax': 'BCFarfield'}, True),
])
def test_lu_match_bit_exact_vs_oracle(n, scheme, model, perturb, bcs, homo):
    """lu_match = 1: the sweeps of LU/invlu_overlap_l.for / invlu_overlap_u.for (terms in the order i, j, k); the
    restatement is bit-exact against invlu_ of the compiled reference (test_oracle_vs_ref.py::
    test_implicit_lu_match_bit_exact)."""
    extra = dict(SINGLE, lu_match=1)
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb, homo=homo)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[75]) == 1 and int(pint[71]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    ro0 = oz.ro.copy()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4


def test_lu_match_blocks_with_joins():
    blocks, nn = (2, 1, 2), (16, 9, 12)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    kw = dict(model='NSLaminar', blocks=blocks, extra_numz=dict(SINGLE, lu_match=1), extra_numb={'ss_iteration': 3},
              temporal_scheme='implicit', time_step=0.02, bcs={'jmin': 'BCWall'})
    to = cases.tgv(nn, 'senseur', **kw)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
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


@pytest.mark.parametrize("dim,n,scheme,model,bcs,perturb,nb_relax,lu_match", [
    (3, (10, 9, 8), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 2, 0),
    (3, (10, 9, 8), 'ausmpred', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 2, 1),
    (3, (10, 9, 8), 'roe', 'NSTurbulent', None, 0.15, 1, 1),
    (2, (14, 12), 'ausmpred', 'NSLaminar', None, 0.15, 2, 0),
    (2, (14, 12), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 1, 1),
    (2, (40, 11), 'senseur', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 2, 1),   # the reference's own use
    (2, (40, 11), 'roe', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 1, 1),
    (2, (14, 12), 'ausmpred', 'Euler', None, 0., 3, 1),
])
def test_ssor_and_lu_match_six_equations_and_two_dimensional(dim, n, scheme, model, bcs, perturb, nb_relax, lu_match):
    """LU-SSOR relaxations and lu_match = 1 with the Spalart-Allmaras model (LU/invlussor*_SA.for, invlu_overlap_*_SA.for:
    a decoupled 6th equation with its own diagonal) and on two-dimensional zones.  Restatement bit-exact against the
    compiled reference on these cases; device bit for bit (5 equations) / 1e-12 (SA: device pow)."""
    from common import wall_distance
    extra = dict(SINGLE, nb_relax=nb_relax, lu_match=lu_match)
    kw = dict(model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs, temporal_scheme='implicit',
              time_step=0.02, perturb=perturb)
    t = cases.tgv(n, scheme, **kw) if dim == 3 else cases.tgv2d(n, scheme, **kw)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    assert int(pint[71]) == nb_relax and int(pint[75]) == lu_match
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 5):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if neq == 6:
            e = rel_err(got, oz.ro, int(pint[41]), neq=6)
            assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
        else:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4


@pytest.mark.parametrize("n,scheme,model,bcs,perturb,init,modulo", [
    ((12, 10, 9), 'ausmpred', 'NSLaminar', None, 0., 1, 50),
    ((10, 9, 8), 'senseur', 'NSLaminar', {'jmin': 'BCWall'}, 0.15, 3, 3),
    ((10, 9, 8), 'roe', 'NSTurbulent', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0., 2, 2),
    ((40, 11, 9), 'senseur', 'Euler', None, 0., 1, 1),
])
def test_selective_frequency_damping(n, scheme, model, bcs, perturb, init, modulo):
    """sfd.for once per implicit step (lhs.cpp:181-185), before the joins and BCs of the last Newton sub-iteration or, on
    a verification step, after the verification slot.  The restatement is bit-exact against sfd_ of the compiled
    reference; the device exponential differs from the host's in the last bits: 1e-12."""
    from common import wall_distance
    extra = dict(SINGLE, sfd=1, sfd_chi=0.5, sfd_delta=0.3, sfd_init_iter=init)
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=bcs,
                  temporal_scheme='implicit', time_step=0.02, perturb=perturb, modulo_verif=modulo)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    assert int(pint[62]) == 1
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    for it in range(1, 7):
        oz.verif = it % modulo == 0 or it == 1
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]), neq=neq)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    FastS.sync_to_host(t)
    rof = FastC.level_buffer(z, '_f')
    assert np.abs(rof).max() > 0.1
    assert rel_err(rof, oz.rof, int(pint[41]), neq=neq).max() < 1e-12


@pytest.mark.parametrize("n,scheme,model,bcs,perturb,init,rk", [
    ((12, 10, 9), 'senseur', 'NSLaminar', None, 0., 1, 3),
    ((10, 9, 8), 'ausmpred', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, 0.15, 3, 3),
    ((40, 11, 9), 'roe', 'Euler', None, 0., 2, 2),
    ((72, 20, 16), 'senseur', 'NSLaminar', None, 0., 1, 3),
])
def test_selective_frequency_damping_explicit_scheme(n, scheme, model, bcs, perturb, init, rk):
    """The explicit loop of the reference runs the condition of lhs.cpp:181-185 too (no verification slot: nitcfg =
    nssiter - 1): the state is filtered after the last stage but one, before its joins and BCs.  The restatement is
    bit-exact against the compiled reference (test_oracle_vs_ref.py); device exponential: 1e-12."""
    extra = dict(SINGLE, sfd=1, sfd_chi=0.5, sfd_delta=0.3, sfd_init_iter=init)
    t = cases.tgv(n, scheme, model=model, extra_numz=extra, bcs=bcs, time_step=0.02, perturb=perturb,
                  extra_numb={'rk': rk} if rk != 3 else None)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[62]) == 1 and int(pint[58]) == rk
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]))
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    FastS.sync_to_host(t)
    rof = FastC.level_buffer(z, '_f')
    assert np.abs(rof).max() > 0.1
    assert rel_err(rof, oz.rof, int(pint[41])).max() < 1e-12


@pytest.mark.parametrize("n,scheme,model,perturb,bcs,numz", [
    ((96, 70, 52), 'senseur', 'NSLaminar', 0., None, {}),
    ((50, 37, 44), 'ausmpred', 'NSLaminar', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {}),
    ((41, 30, 1), 'roe', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {}),
    ((44, 37, 30), 'senseur', 'NSLaminar', 0.15, {'imin': 'BCWall', 'imax': 'BCFarfield'}, {'nb_relax': 2}),
    ((41, 30, 1), 'roe', 'NSTurbulent', 0.15, {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {'nb_relax': 2, 'lu_match': 1}),
    ((33, 26, 20), 'ausmpred', 'NSTurbulent', 0., {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {'lu_match': 1}),
])
def test_sweeps_in_one_launch_equal_one_launch_per_hyperplane(n, scheme, model, perturb, bcs, numz, monkeypatch):
    """The LU-SGS sweeps as one launch each (pencils of 8 lines that wait for each other's step counts) against one
    launch per hyperplane, bit for bit, on zones with many pencils in work at a time, with ragged line and pencil counts;
    with 1, 5 and 64 waves the order in which the pencils are handed out is all that keeps a wave from waiting for
    a pencil nobody has taken.  LU-SGS, LU-SSOR relaxations and lu_match = 1."""
    def run(flow, waves):
        monkeypatch.setenv('FASTS_HIP_LU_FLOW', str(flow))
        monkeypatch.setenv('FASTS_HIP_LU_FLOW_WAVES', str(waves))
        if len(n) == 3 and n[2] == 1:
            t = cases.tgv2d(n[:2], scheme, model=model, extra_numz=dict(SINGLE, **numz), bcs=bcs, perturb=perturb, time_step=0.02,
                            temporal_scheme='implicit', extra_numb={'ss_iteration': 3, 'modulo_verif': 1})
        else:
            t = cases.tgv(n, scheme, model=model, extra_numz=dict(SINGLE, **numz), extra_numb={'ss_iteration': 3}, bcs=bcs,
                          temporal_scheme='implicit', time_step=0.02, perturb=perturb, modulo_verif=1)
        (t, tc, metrics) = FastS.warmup(t, None)
        z = Internal.getZones(t)[0]
        for it in range(1, 4):
            FastS._compute(t, metrics, it, None)
        out = FastC.level_buffer(z).copy()
        FastC.HOOK['ctx'].close()
        return out
    ref = run(0, 768)
    assert np.isfinite(ref).all()
    for waves in (768, 64, 5, 1):
        got = run(1, waves)
        assert np.array_equal(got, ref), "%d waves: max abs diff %g" % (waves, np.abs(got - ref).max())
