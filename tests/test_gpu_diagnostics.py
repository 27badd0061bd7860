"""Taylor-Green diagnostics on the device (kinetic energy and enstrophy, C ABI fs_tgv_sums) against the CPU restatement
and against post_ of the compiled reference (golden vectors)."""
import os
import sys
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import SINGLE

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG


@pytest.mark.parametrize("n,scheme", [(16, 'senseur'), ((9, 14, 11), 'ausmpred'), ((70, 10, 9), 'senseur')])
def test_tgv_sums_vs_oracle(n, scheme):
    t = cases.tgv(n, scheme, extra_numz=SINGLE)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
    preal = Internal.getNodeFromName2(z, 'Parameter_real')[1]
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
    ctx = FastC.HOOK['ctx']
    for it in range(1, 4):
        FastS._compute(t, metrics, it, None)
        oz.step()
        assert np.array_equal(FastC.level_buffer(z), oz.ro)
        got = ctx.tgv_sums()
        np.testing.assert_allclose(got, oz.tgv_sums(), rtol=1e-12)
    enst, tke = FastS._computeEnstrophy(t, metrics)
    assert enst == got[0] / got[2] and tke == got[1] / got[2]
    assert 0.1 < tke < 0.2          # 0.125 rho0 U0^2 for the Taylor-Green field


@pytest.mark.parametrize("name", ['tgv12_senseur_lamin', 'tgv_2x2_senseur_lamin'])
def test_tgv_sums_vs_reference_golden(name):
    """Final state of a golden case: the device sums against post_ of the compiled reference, zones added up."""
    spec = MG.CASES[name]
    kind, n, scheme, model, blocks = spec[:5]
    g = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    t = MG.build(kind, n, scheme, model, blocks, 'single')
    tc = None
    zones = Internal.getZones(t)
    joins = MG.case_joins(kind, n, blocks)
    if joins:
        tc = C.node2Center(t)
        connect.add_joins_to_tc(tc, joins, [z[0] for z in zones])
    (t, tc, metrics) = FastS.warmup(t, tc)
    for it in range(1, int(g['nit']) + 1):
        FastS._compute(t, metrics, it, tc)
    got = FastC.HOOK['ctx'].tgv_sums()
    np.testing.assert_allclose(got, g['tgvsums'].sum(axis=0), rtol=1e-12)
