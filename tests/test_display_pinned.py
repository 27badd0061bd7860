"""display_temporal_criteria against the reference's own display routines: `dpssiter_` (the printed table of the
Newton sub-iterations) and `conv2pytree_` (the ZoneConvergenceHistory arrays) of the compiled reference were run on
the implicit golden cases by tests/golden/make_golden.py; their output is in the golden files (`disp_text`,
`disp_text_long_all`, `zch_*`).  CPU part: the formatting functions of fast_amd.PyTree fed with the reference's norms
reproduce the text character for character and the history to the last bits.  GPU part: the device run, through
display_temporal_criteria and a ZoneConvergenceHistory node, gives the same table and history."""
import os
import re
import sys
import numpy as np
import pytest

import fast_amd.PyTree as FastS
from fast_amd import connect, FastC, Internal
import fast_amd.pytree_lite as C

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG

IMPLICIT = sorted(k for k, v in MG.CASES.items() if v[5] == 'single' and
                  os.path.exists(os.path.join(HERE, 'golden', k + '.npz')) and
                  'disp_text' in np.load(os.path.join(HERE, 'golden', k + '.npz')).files)


def test_there_are_pinned_cases():
    assert len(IMPLICIT) >= 2


@pytest.mark.parametrize("name", IMPLICIT)
def test_formatting_reproduces_the_reference_text_and_history(name):
    g = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    hist, names = g['reshist'], [str(s) for s in g['zone_names']]
    nit, nb, nss, _, neq = hist.shape
    for it in range(1, nit + 1):
        lines = []
        for b in range(nb):
            lines += FastS.residual_table_lines(names[b], it, nss, True, hist[it - 1, b, :, 0], hist[it - 1, b, :, 1], 1, 0)
        assert '\n'.join(lines) + '\n' == str(g['disp_text'][it - 1]), "step %d" % it
    lines = []
    for b in range(nb):
        lines += FastS.residual_table_lines(names[b], nit, nss, True, hist[nit - 1, b, :, 0], hist[nit - 1, b, :, 1], 0, 2)
    assert '\n'.join(lines) + '\n' == str(g['disp_text_long_all'])
    for b in range(nb):
        assert int(g['zch_last_%d' % b][0]) == nit
        assert np.array_equal(g['zch_itnum_%d' % b][:nit], np.arange(1, nit + 1))
        for it in range(1, nit + 1):
            r = FastS.convergence_record(hist[it - 1, b, :, 0], hist[it - 1, b, :, 1])
            sl = slice((it - 1) * neq, it * neq)
            for got, key in zip(r, ('l2', 'oo', 'l2_diff', 'oo_diff')):
                np.testing.assert_allclose(got, g['zch_%s_%d' % (key, b)][sl], rtol=0, atol=1e-13)


def _numbers(line):
    return [float(x) for x in re.findall(r'-?\d+\.\d+', line.split(')', 1)[1])]


@pytest.mark.gpu
@pytest.mark.parametrize("name", IMPLICIT)
def test_device_history_and_table_equal_the_reference(name, capsys):
    spec = MG.CASES[name]
    kind, n, scheme, model, blocks = spec[:5]
    g = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    t = MG.build(kind, n, scheme, model, blocks, 'single', spec[7] if len(spec) > 7 else None,
                 spec[8] if len(spec) > 8 else None, modulo_verif=1)
    tc = None
    joins = MG.case_joins(kind, n, blocks)
    if joins:
        tc = C.node2Center(t)
        connect.add_joins_to_tc(tc, joins, [z[0] for z in Internal.getZones(t)])
    (t, tc, metrics) = FastS.warmup(t, tc)
    nit = int(g['nit'])
    FastS._createConvergenceHistory(t, nit + 2)
    zones = Internal.getZones(t)
    neq = int(g['reshist'].shape[4])
    # Spalart-Allmaras: the device pow of the destruction term moves the residuals by 1e-12 relative
    tol = 1e-9 if model == 'NSTurbulent' else 1e-10
    for it in range(1, nit + 1):
        FastS._compute(t, metrics, it, tc)
        capsys.readouterr()
        FastS.display_temporal_criteria(t, metrics, it, format='single')
        got = [ln for ln in capsys.readouterr().out.splitlines() if not ln.startswith('CFL_MOY_MAX_MIN')]
        ref = str(g['disp_text'][it - 1]).splitlines()
        assert len(got) == len(ref), "step %d:\n%s\n--- reference:\n%s" % (it, '\n'.join(got), '\n'.join(ref))
        for a, b in zip(got, ref):
            if a != b:       # a value that sits on a rounding boundary of the two decimals
                assert a.split(')', 1)[0] == b.split(')', 1)[0]
                assert np.allclose(_numbers(a), _numbers(b), rtol=0, atol=0.0101), "step %d: %r / %r" % (it, a, b)
    # the long format with every sub-iteration, after the last step
    capsys.readouterr()
    for z in zones:     # the call above has stored a record for the step: take it back
        zch = Internal.getNodeFromName1(z, 'ZoneConvergenceHistory')
        assert int(np.asarray(zch[1]).ravel()[0]) == nit
    hold = [np.array(Internal.getNodeFromName1(Internal.getNodeFromName1(z, 'ZoneConvergenceHistory'), 'RSD_L2')[1], copy=True)
            for z in zones]
    FastS.display_temporal_criteria(t, metrics, nit, format='double', verbose='all')
    got = [ln for ln in capsys.readouterr().out.splitlines() if not ln.startswith('CFL_MOY_MAX_MIN')]
    ref = str(g['disp_text_long_all']).splitlines()
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert a.split(')', 1)[0] == b.split(')', 1)[0]
        np.testing.assert_allclose(_numbers(a), _numbers(b), rtol=0, atol=2e-9)
    for b, z in enumerate(zones):
        zch = Internal.getNodeFromName1(z, 'ZoneConvergenceHistory')
        assert np.array_equal(Internal.getNodeFromName1(zch, 'IterationNumber')[1][:nit], g['zch_itnum_%d' % b][:nit])
        for key, node in (('l2', 'RSD_L2'), ('oo', 'RSD_oo'), ('l2_diff', 'RSD_L2_diff'), ('oo_diff', 'RSD_oo_diff')):
            arr = hold[b] if key == 'l2' else Internal.getNodeFromName1(zch, node)[1]
            np.testing.assert_allclose(arr[:nit * neq], g['zch_%s_%d' % (key, b)][:nit * neq], rtol=0, atol=tol,
                                       err_msg="%s of zone %d" % (node, b))
