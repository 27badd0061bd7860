"""GPU parity of the boundary conditions that read window data from Parameter_real: pressure outlet
(BCOutpres, tag 16) and isothermal no-slip wall (BCWallViscousIsothermal, tag 32), against the CPU restatement,
itself bit-exact against bvbs_outpres / bvbs_wall_viscous_isothermal of the compiled reference
(test_oracle_vs_ref.py).  Mean-flow cases must agree bit for bit; the Spalart-Allmaras ones to 1e-12 (the device
pow of the destruction term differs from the host pow in the last bit)."""
import numpy as np
import pytest

from fast_amd import cases, FastC, Internal, _capi
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, rel_err, SINGLE, wall_distance, data_bcs, window_data

pytestmark = pytest.mark.gpu

CASES = [
    # n, model, scheme, perturb, temporal, with the injection BC (its Newton takes a power: 1e-12 instead of bit for bit)
    ((9, 8, 7), 'NSLaminar', 'ausmpred', 0., 'explicit', False),
    ((70, 9, 12), 'NSLaminar', 'senseur', 0., 'explicit', False),
    ((10, 9, 8), 'Euler', 'senseur', 0., 'explicit', False),
    ((10, 8, 9), 'NSLaminar', 'roe', 0.15, 'explicit', False),
    ((9, 8, 10), 'NSTurbulent', 'ausmpred', 0.15, 'explicit', True),
    ((10, 9, 8), 'NSLaminar', 'senseur', 0.15, 'implicit', False),
    ((9, 10, 8), 'NSTurbulent', 'ausmpred', 0., 'implicit', False),
    ((34, 9, 8), 'NSLaminar', 'ausmpred', 0., 'explicit', True),
    ((10, 8, 9), 'Euler', 'roe', 0.15, 'explicit', True),
]


@pytest.mark.parametrize("n,model,scheme,perturb,temporal,inj", CASES)
def test_bcs_with_window_data(n, model, scheme, perturb, temporal, inj):
    bcs = data_bcs(n, model)
    implicit = temporal == 'implicit'
    if not inj:
        del bcs['jmax']
    if implicit:
        bcs['jmax'] = 'BCWall'
    t = cases.tgv(n, scheme, model=model, extra_numz=SINGLE, extra_numb={'ss_iteration': 3}, bcs=bcs, perturb=perturb,
                  temporal_scheme=temporal, time_step=0.02 if implicit else None)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    if inj:
        assert rel_err(FastC.level_buffer(z), oz.ro, int(pint[41]), neq=neq).max() < 1e-14
    else:
        assert np.array_equal(FastC.level_buffer(z), oz.ro)    # ghost fill of the initial field: bit-exact
    if implicit:
        oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        if neq == 6 or inj:
            e = rel_err(got, oz.ro, int(pint[41]), neq=neq)
            assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
        else:
            assert np.array_equal(got, oz.ro), "step %d: max rel err %s" % (it, rel_err(got, oz.ro, int(pint[41])))
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-6


def test_window_data_must_match_the_window():
    n = (8, 8, 8)
    t = cases.tgv(n, 'senseur', extra_numz=SINGLE, bcs={'imax': ('BCOutpres', {'Pressure': window_data((5, 5), 70.)})})
    with pytest.raises(_capi.FastsHipError, match="window data"):
        FastS.warmup(t, None)
    t = cases.tgv(n, 'senseur', extra_numz=SINGLE, bcs={'imax': 'BCOutpres'})
    with pytest.raises(_capi.FastsHipError, match="window data"):
        FastS.warmup(t, None)


@pytest.mark.parametrize("model,scheme,n,perturb,temporal", [
    ('NSLaminar', 'ausmpred', (70, 9, 12), 0., 'explicit'),
    ('Euler', 'roe', (10, 9, 8), 0., 'explicit'),
    ('NSLaminar', 'senseur', (10, 9, 8), 0.15, 'explicit'),
    ('NSTurbulent', 'ausmpred', (10, 9, 8), 0.15, 'explicit'),
    ('NSLaminar', 'ausmpred', (10, 9, 8), 0.15, 'implicit'),
])
def test_mass_flow_rate_bcs(model, scheme, n, perturb, temporal):
    """BCOutMFR (bvbs_outmfr.for) and BCInjMFR (bvbs_injmfr.for, the quadratic and the Newton start): device against the
    CPU restatement, itself bit-exact against the compiled reference.  1e-12: the Newton start raises to the power
    gamma/(gamma-1), and the device pow is not the host's."""
    from fast_amd import cases, FastC, Internal
    import fast_amd.PyTree as FastS
    from oracle import oracle as O
    from common import SINGLE, mfr_bcs, zone_arrays, rel_err, wall_distance
    bcs = mfr_bcs(n, model)
    if model == 'Euler':
        bcs['jmin'] = 'BCWallInviscid'
    t = cases.tgv(n, scheme, model=model, extra_numz=SINGLE, bcs=bcs, perturb=perturb,
                  temporal_scheme=temporal, time_step=0.02 if temporal == 'implicit' else None)
    (t, tc, metrics) = FastS.warmup(t, None)
    z = Internal.getZones(t)[0]
    pint, preal, _ = zone_arrays(z)
    neq = int(pint[36])
    oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0], wall_distance(z) if neq == 6 else None)
    oz.fill_ghostcells()
    assert rel_err(FastC.level_buffer(z), oz.ro, int(pint[41]), neq=neq).max() < 1e-12
    oz.ro_m1[:] = oz.ro; oz.ro_p1[:] = oz.ro
    ro0 = oz.ro.copy()
    for it in range(1, 6):
        FastS._compute(t, metrics, it, None)
        oz.step()
        got = FastC.level_buffer(z)
        e = rel_err(got, oz.ro, int(pint[41]), neq=neq)
        assert e.max() < 1e-12, "step %d: rel err %s" % (it, e)
    assert np.isfinite(got).all() and np.abs(got - ro0).max() > 1e-5
