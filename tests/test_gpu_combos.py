"""Option combinations that no single-feature test crosses: k-homogeneous zones with a local time step and with
lu_match, selective frequency damping with joins and with LU-SSOR relaxations, SSOR on blocks with joins.  Device
against the CPU restatement, bit for bit (1e-12 where the device exponential or power enters)."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare, wall_distance

pytestmark = pytest.mark.gpu

WALLS = {'jmin': 'BCWall', 'jmax': 'BCFarfield'}


def _single(t, nit, tol=0., verif_every=None):
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, nit + 1):
        if verif_every:
            oz.verif = it % verif_every == 0 or it == 1
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if tol:
            e = rel_err(got, oz.ro, int(pint[41]), neq=neq)
            assert e.max() < tol, "step %d: rel err %s" % (it, e)
        else:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41]), neq=neq))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-4
    return pint


def test_homogeneous_zone_implicit_local_time_step_lu_match():
    extra = dict(SINGLE, time_step_nature='local', cfl=4., lu_match=1)
    t = cases.tgv((12, 10, 9), 'senseur', extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=WALLS,
                  temporal_scheme='implicit', time_step=0.02, perturb=0.2, homo=True)
    pint = _single(t, 4)
    assert int(pint[25]) == 1 and int(pint[46]) == 1 and int(pint[75]) == 1


def test_homogeneous_zone_explicit_local_time_step():
    extra = dict(SINGLE, time_step_nature='local', cfl=0.5)
    t = cases.tgv((12, 10, 9), 'ausmpred', extra_numz=extra, bcs=WALLS, perturb=0.2, homo=True)
    pint = _single(t, 4)
    assert int(pint[25]) == 1 and int(pint[46]) == 1


def test_frequency_damping_with_ssor_and_spalart_allmaras():
    extra = dict(SINGLE, sfd=1, sfd_chi=0.4, sfd_delta=0.5, sfd_init_iter=2, nb_relax=2, lu_match=1)
    t = cases.tgv((10, 9, 8), 'senseur', model='NSTurbulent', extra_numz=extra, extra_numb={'ss_iteration': 3}, bcs=WALLS,
                  temporal_scheme='implicit', time_step=0.02, perturb=0.15, modulo_verif=2)
    pint = _single(t, 5, tol=1e-12, verif_every=2)
    assert int(pint[62]) == 1 and int(pint[71]) == 2 and int(pint[36]) == 6


@pytest.mark.parametrize("opts", [dict(nb_relax=2), dict(sfd=1, sfd_chi=0.5, sfd_delta=0.3), dict(lu_match=1, nb_relax=2)])
def test_implicit_options_on_blocks_with_joins(opts):
    blocks, nn = (2, 1, 2), (16, 9, 12)
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    kw = dict(model='NSLaminar', blocks=blocks, extra_numz=dict(SINGLE, **opts), extra_numb={'ss_iteration': 3},
              temporal_scheme='implicit', time_step=0.02, bcs={'jmin': 'BCWall'}, perturb=0.15)
    to = cases.tgv(nn, 'ausmpred', **kw)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in joins])
    s.fill_ghostcells()
    t = cases.tgv(nn, 'ausmpred', **kw)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    tol = 1e-12 if 'sfd' in opts else 0.
    for it in range(1, 5):
        for oz in s.zones:
            oz.verif = it == 1          # modulo_verif = 50
        FastS._compute(t, metrics, it, tc)
        s.step()
        for z, oz in zip(zones, s.zones):
            got = FastC.level_buffer(z)
            if tol:
                assert rel_err(got, oz.ro, int(oz.pint[41])).max() < tol, "step %d zone %s" % (it, z[0])
            else:
                assert np.array_equal(got, oz.ro), "step %d zone %s" % (it, z[0])
