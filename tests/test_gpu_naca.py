"""GPU parity on the steady-RANS set-up of a two-dimensional NACA0012 C-mesh: Spalart-Allmaras, implicit
LU-SGS with a local time step, wall on the section, far field outside, match join across the wake cut (the
zone is its own donor).  The CPU restatement of this case is bit-exact against the compiled reference
(test_oracle_vs_ref.py::test_naca0012_cmesh_bit_exact)."""
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, prepare, wall_distance

pytestmark = pytest.mark.gpu


def _setup(nf, nw, nj, **kw):
    t = cases.naca_cmesh(nfoil=nf, nwake=nw, nj=nj, extra_numz=dict(SINGLE, **kw.pop('numz', {})), **kw)
    rp, dp = connect.cmesh_wake_join(2 * nf + 2 * nw, nj, nw)
    return t, [(0, 0, rp, dp)]


@pytest.mark.parametrize("scheme,model,temporal,nf,nw,nj", [
    ('ausmpred', 'NSTurbulent', 'implicit', 16, 6, 10),
    ('senseur', 'NSTurbulent', 'implicit', 24, 9, 12),
    ('ausmpred', 'NSLaminar', 'implicit', 16, 6, 10),
    ('ausmpred', 'NSTurbulent', 'explicit', 16, 6, 10),
])
def test_naca0012_cmesh_matches_oracle(scheme, model, temporal, nf, nw, nj):
    kw = dict(scheme=scheme, model=model, temporal_scheme=temporal)
    if temporal == 'explicit':
        kw['numz'] = {'time_step': 2.e-5}
    to, joins = _setup(nf, nw, nj, **dict(kw))
    m = prepare(to)
    zo = Internal.getZones(to)[0]
    p, pr, _ = zone_arrays(zo)
    oz = O.OracleZone(p, pr, FastC.level_buffer(zo), m[0][0], wall_distance(zo))
    s = O.OracleSolver([oz], [(br, r, bd, d) for br, bd, r, d in joins])
    s.fill_ghostcells()
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro

    t, joins = _setup(nf, nw, nj, **dict(kw))
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    z = Internal.getZones(t)[0]
    neq = oz.neq
    tol = 1e-12 if neq == 6 else 0.
    assert rel_err(FastC.level_buffer(z), oz.ro, oz.ndimdx, neq=neq).max() <= tol
    ro0 = oz.ro.copy()
    for it in range(1, 9):
        FastS._compute(t, metrics, it, tc)
        s.step()
        e = rel_err(FastC.level_buffer(z), oz.ro, oz.ndimdx, neq=neq)
        assert e.max() <= tol, "step %d: %s" % (it, e)
    assert np.isfinite(oz.ro).all() and np.abs(oz.ro - ro0).max() > 1e-4
