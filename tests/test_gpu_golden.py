"""GPU path (through the C ABI) against the golden vectors produced by the compiled reference:
bit-exact fields after 1, 3 and nit time steps, residual history to summation order."""
import glob
import os
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}

CASES = {
    'tgv12_senseur_lamin':      ('tgv', 12, 'senseur', 'NSLaminar', (1, 1, 1)),
    'tgv12_ausmpred_lamin':     ('tgv', 12, 'ausmpred', 'NSLaminar', (1, 1, 1)),
    'tgv12_roe_lamin':          ('tgv', 12, 'roe', 'NSLaminar', (1, 1, 1)),
    'tgv10_senseur_euler':      ('tgv', 10, 'senseur', 'Euler', (1, 1, 1)),
    'vortex12_roe_euler':       ('vortex', 12, 'roe', None, (1, 1, 1)),
    'tgv_ragged_senseur_lamin': ('tgv', (13, 6, 9), 'senseur', 'NSLaminar', (1, 1, 1)),
    'tgv_2x2_senseur_lamin':    ('tgv', (8, 12, 20), 'senseur', 'NSLaminar', (1, 2, 2)),
    'channel_senseur_lamin':    ('tgv', (12, 10, 8), 'senseur', 'NSLaminar', (1, 1, 1),
                                 {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}),
    'box_farfield_ausmpred':    ('tgv', (9, 8, 10), 'ausmpred', 'NSLaminar', (1, 1, 1),
                                 {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}),
    'box_mixed_bc_euler':       ('tgv', (8, 9, 7), 'ausmpred', 'Euler', (1, 1, 1),
                                 {'imin': 'BCExtrapolate', 'imax': 'BCFarfield', 'jmin': 'BCWall',
                                  'jmax': 'BCSymmetryPlane', 'kmin': 'BCWallInviscid'}),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_equals_reference_golden(name):
    kind, n, scheme, model, blocks = CASES[name][:5]
    bcs = CASES[name][5] if len(CASES[name]) > 5 else None
    g = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    if kind == 'tgv':
        t = cases.tgv(n, scheme, model=model, blocks=blocks, extra_numz=SINGLE, modulo_verif=1, bcs=bcs)
    else:
        t = cases.vortex(n, scheme, extra_numz=SINGLE)
        FastC._setNum2Base(t, {'temporal_scheme': 'explicit', 'modulo_verif': 1})
    tc = None
    zones = Internal.getZones(t)
    if len(zones) > 1:
        nv = tuple(n[d] // blocks[d] for d in range(3))
        tc = C.node2Center(t)
        connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, nv), [z[0] for z in zones])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for b, z in enumerate(zones):
        # the inputs the reference was run on are the inputs the device got
        assert np.array_equal(Internal.getNodeFromName2(z, 'Parameter_int')[1], g['pint%d' % b])
        assert np.array_equal(metrics[b][0], g['mtr%d' % b])
        assert np.array_equal(FastC.level_buffer(z), g['ghost_%d' % b]), "ghost fill of zone %d" % b
    nit = int(g['nit'])
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, tc)
        res = FastS.display_temporal_criteria(t, metrics, it, format='none')
        for b, z in enumerate(zones):
            l2, loo = res[z[0]]
            np.testing.assert_allclose(l2, g['residuals'][it - 1, b, 0], rtol=1e-11, atol=1e-300)
            np.testing.assert_allclose(loo, g['residuals'][it - 1, b, 1], rtol=1e-12, atol=1e-300)
            key = 'ro_it%d_%d' % (it, b)
            if key in g.files:
                got = FastC.level_buffer(z)
                assert np.array_equal(got, g[key]), "step %d zone %d: max abs diff %g" % (
                    it, b, np.abs(got - g[key]).max())
