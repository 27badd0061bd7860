"""The CPU restatement on OpenMP threads (k-slabs of the zone, fo_subiter_mt) gives what its one-thread sweep gives,
bit for bit: each cell receives its flux contributions in the same order."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal
from oracle import oracle as O
from common import prepare, zone_arrays, wall_distance


def _run(n, scheme, model, threads, nit, kw):
    t = cases.tgv(n, scheme, model=model, **kw)
    m = prepare(t)
    z = Internal.getZones(t)[0]
    p, pr, _ = zone_arrays(z)
    oz = O.OracleZone(p, pr, FastC.level_buffer(z), m[0][0], wall_distance(z) if model == 'NSTurbulent' else None)
    oz.threads = threads
    oz.fill_ghostcells()
    for it in range(nit):
        oz.step()
    return oz.ro.copy()


@pytest.mark.parametrize("n,scheme,model,kw", [
    ((20, 14, 24), 'senseur', 'NSLaminar', {}),
    ((20, 14, 23), 'ausmpred', 'NSLaminar', {'perturb': 0.15}),
    ((16, 12, 20), 'roe', 'NSLaminar', {'bcs': {'kmin': 'BCWall', 'kmax': 'BCWall', 'jmin': 'BCWall'}}),
    ((16, 12, 20), 'senseur', 'NSTurbulent', {'bcs': {'jmin': 'BCWall'}}),
    ((12, 10, 7), 'senseur', 'Euler', {}),          # fewer planes than two per thread: one thread
])
@pytest.mark.parametrize("threads", [3, 4])
def test_threads_equal_one_thread(n, scheme, model, kw, threads):
    a = _run(n, scheme, model, 1, 3, kw)
    b = _run(n, scheme, model, threads, 3, kw)
    assert np.isfinite(a).all()
    assert np.array_equal(a, b)
