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
