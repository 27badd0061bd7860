"""GPU parity on curvilinear zones (zone type 0: surface vectors and volumes stored per cell):
stage kernel, boundary conditions whose normals come from the face metrics, joins, residual norms.
Checked bit for bit against the CPU restatement, itself bit-exact against the compiled reference
on the same family of cases (test_oracle_vs_ref.py)."""
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
    # n, scheme, model, bcs, steps
    ((14, 12, 10), 'senseur', 'NSLaminar', None, 6),
    ((12, 10, 9), 'ausmpred', 'NSLaminar', None, 5),
    ((10, 9, 12), 'roe', 'NSLaminar', None, 5),
    ((12, 9, 10), 'senseur', 'Euler', None, 5),
    ((9, 12, 10), 'ausmpred', 'Euler', None, 4),
    ((10, 10, 9), 'roe', 'Euler', None, 4),
    ((70, 17, 37), 'senseur', 'NSLaminar', None, 3),       # several patches in i and j, two k chunks
    ((31, 7, 33), 'ausmpred', 'NSLaminar', None, 3),       # exactly one patch of owned cells
    ((32, 8, 5), 'senseur', 'NSLaminar', None, 3),         # one cell more than a patch
    ((12, 10, 9), 'senseur', 'NSLaminar', {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}, 5),
    ((10, 9, 8), 'ausmpred', 'Euler', {'kmin': 'BCWallInviscid', 'kmax': 'BCWallInviscid', 'imax': 'BCSymmetryPlane'}, 5),
    ((9, 8, 10), 'senseur', 'NSLaminar', FAR, 5),
    ((8, 9, 7), 'roe', 'Euler', {'imin': 'BCExtrapolate', 'imax': 'BCFarfield', 'kmin': 'BCExtrapolate'}, 5),
]


@pytest.mark.parametrize("n,scheme,model,bcs,nit", CASES)
def test_curvilinear_bit_exact_vs_oracle(n, scheme, model, bcs, nit):
    t = cases.tgv(n, scheme, model=model, extra_numz=SINGLE, bcs=bcs, perturb=0.15)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    assert int(pint[25]) == 0
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    oz.fill_ghostcells()
    assert np.array_equal(FastC.level_buffer(z), oz.ro)
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all()
    # residual norms of the first stage use dt/vol of each cell
    l2, loo = FastS.display_temporal_criteria(t, metrics, 1, format='none')[z[0]]
    assert np.isfinite(l2).all() and np.isfinite(loo).all()


def test_curvilinear_residuals_match_oracle():
    t = cases.tgv((12, 10, 9), 'senseur', extra_numz=SINGLE, perturb=0.2, modulo_verif=1)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    FastS._compute(t, metrics, 1, None)
    oz.step()
    l2, loo = FastS.display_temporal_criteria(t, metrics, 1, format='none')[z[0]]
    rl2, rloo = oz.residuals()
    np.testing.assert_allclose(l2, rl2, rtol=1e-12)
    np.testing.assert_allclose(loo, rloo, rtol=1e-13)


@pytest.mark.parametrize("blocks,nn,scheme,model", [
    ((2, 1, 2), (16, 9, 12), 'senseur', 'NSLaminar'),
    ((1, 2, 1), (10, 16, 8), 'ausmpred', 'Euler'),
])
def test_curvilinear_blocks_with_joins(blocks, nn, scheme, model):
    nv = tuple(nn[d] // blocks[d] for d in range(3))
    joins = connect.periodic_block_joins(blocks, nv)
    to = cases.tgv(nn, scheme, model=model, blocks=blocks, extra_numz=SINGLE, perturb=0.15)
    m = prepare(to)
    zs = []
    for z, mm in zip(Internal.getZones(to), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in joins])
    s.fill_ghostcells()
    t = cases.tgv(nn, scheme, model=model, blocks=blocks, extra_numz=SINGLE, perturb=0.15)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for z, oz in zip(zones, s.zones):
        assert np.array_equal(FastC.level_buffer(z), oz.ro)
    for it in range(1, 4):
        FastS._compute(t, metrics, it, tc)
        s.step()
        for z, oz in zip(zones, s.zones):
            assert np.array_equal(FastC.level_buffer(z), oz.ro), "step %d zone %s" % (it, z[0])
