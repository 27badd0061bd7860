"""Shared helpers of the test-suite: build a case, prepare its solver state, wrap the oracle."""
import numpy as np
from fast_amd import cases, FastC, Internal


SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def prepare(t):
    """Host-side preparation without touching the GPU (what warmup does before creating the
    device context)."""
    FastC._buildOwnData(t, None)
    FastC._build_omp(t)
    dtloc = Internal.getNodeFromName2(t, '.Solver#dtloc')[1]
    zones = Internal.getZones(t)
    metrics = [FastC.allocate_metric_zone(z, int(dtloc[0])) for z in zones]
    FastC._BCcompact(t)
    FastC.fill_task_records(dtloc, zones)
    FastC._createPrimVars(t)
    return metrics


def zone_arrays(z):
    pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
    preal = Internal.getNodeFromName2(z, 'Parameter_real')[1]
    g = Internal.getNodeFromName1(z, 'GridCoordinates')
    xyz = [Internal.getNodeFromName1(g, c)[1] for c in ('CoordinateX', 'CoordinateY', 'CoordinateZ')]
    return pint, preal, xyz


def make_case(kind, n, scheme, model=None, single_block=True, **kw):
    extra = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000} \
        if single_block else None
    if kind == 'tgv':
        return cases.tgv(n, scheme, model=model or 'NSLaminar', extra_numz=extra, **kw)
    return cases.vortex(n, scheme, extra_numz=extra, **kw)


def rel_err(a, b, ndimdx, neq=5):
    """max |a-b| per variable, scaled by the variable's magnitude (velocities share one scale)."""
    a = a.reshape(neq, -1)[:, :ndimdx]; b = b.reshape(neq, -1)[:, :ndimdx]
    d = np.abs(a - b).max(axis=1)
    s = np.abs(b).max(axis=1)
    vs = s[1:4].max()
    s[1:4] = vs if vs > 0 else 1.
    s[s == 0] = 1.
    return d / s


def wall_distance(z):
    """TurbulentDistance of a zone (Fortran-ordered cell array) or None."""
    sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
    n = Internal.getNodeFromName1(sol, 'TurbulentDistance') if sol is not None else None
    return None if n is None else n[1]


def window_data(shape, base, amp=0.02):
    """Deterministic window data for BCs that carry their own (one value per face cell)."""
    import numpy as _np
    n = int(_np.prod(shape))
    return base * (1. + amp * _np.sin(0.7 * _np.arange(n) + 0.3))


def numpy_full(n, v):
    import numpy as _np
    return _np.full(n, v)


def data_bcs(n, model='NSLaminar', inj=True):
    """BCOutpres on imax and kmin, isothermal wall on jmin (laminar / turbulent only), with window data."""
    pinf = 1. / (1.4 * 0.1 * 0.1)    # adim2funk at M = 0.1: rho = T = 1
    ni, nj, nk = n
    bcs = {'imax': ('BCOutpres', {'Pressure': window_data((nj, nk), pinf)}),
           'kmin': ('BCOutpres', {'Pressure': window_data((ni, nj), pinf, 0.01)}),
           'imin': 'BCFarfield'}
    neq = 6 if model == 'NSTurbulent' else 5
    with_inj = inj
    # supersonic inflow with its own conservative state (rho, rho u, rho v, rho w, rho E [, rho nu_tilde])
    nf = ni * nj
    ro = window_data((nf,), 1.02); u = window_data((nf,), 0.11, 0.1)
    cv = 1. / (1.4 * 0.4 * 1.4 * 0.01)    # adim2funk: cv = 1 / (gamma (gamma-1) M^2)
    sup = {'Density': ro, 'MomentumX': ro * 0.02, 'MomentumY': ro * 0.01, 'MomentumZ': -ro * u,
           'EnergyStagnationDensity': ro * (cv * 1.01 + 0.5 * (0.02 ** 2 + 0.01 ** 2 + u * u))}
    if neq == 6:
        sup['TurbulentSANuTildeDensity'] = ro * window_data((nf,), 2e-5, 0.3)
    bcs['kmax'] = ('BCInflowSupersonic', sup)
    # subsonic injection: direction, stagnation pressure and stagnation enthalpy (BCInj1), on jmax
    ng = ni * nk
    hinf = 1.4 * cv * 1. + 0.5 * 1.     # cp T + u^2/2 of the reference state (rho = T = u = 1)
    inj = {'d0x': numpy_full(ng, 0.1), 'd0y': numpy_full(ng, -0.99), 'd0z': window_data((ng,), 0.05, 0.5),
           'StagnationPressure': window_data((ng,), pinf * 1.01, 0.002),
           'StagnationEnthalpy': window_data((ng,), hinf, 0.001)}
    if neq == 6:
        inj['TurbulentSANuTilde'] = window_data((ng,), 2e-5, 0.3)
    if with_inj:
        bcs['jmax'] = ('BCInj1', inj)
    if model != 'Euler':
        bcs['jmin'] = ('BCWallViscousIsothermal', {'Temperature': window_data((ni, nk), 1.05)})
    return bcs


def own_state_bcs():
    """Far field, inflow and supersonic inflow that carry their own uniform conservative state (rho, rho u, rho v,
    rho w, rho E, rho nu_tilde): one array of 6 values, or the 21 scalars of a ReferenceState copy."""
    import numpy as _np
    cv = 1. / (1.4 * 0.4 * 1.4 * 0.01)
    def state(ro, u, v, w, T, nut):
        return _np.array([ro, ro * u, ro * v, ro * w, ro * (cv * T + 0.5 * (u * u + v * v + w * w)), ro * nut])
    ref21 = _np.arange(21) * 0.
    ref21[3:8] = state(0.97, 1.1, -0.05, 0.02, 1.02, 0.)[:5]; ref21[16] = 0.97 * 3e-5
    return {'imin': ('BCInflow', {'State': state(1.03, 0.9, 0.1, 0., 0.99, 2e-5)}),
            'imax': ('BCFarfield', {'State': state(0.98, 1.05, 0., 0.05, 1.01, 4e-5)}),
            'kmax': ('BCInflowSupersonic', {'State': state(1.01, 0.2, 0.1, -1.2, 1., 1e-5)}),
            'jmax': ('BCFarfield', {'s%02d' % i: _np.array([ref21[i]]) for i in range(21)}),
            'jmin': 'BCWall'}


def mfr_bcs(n, model='NSLaminar', rest_face=True):
    """Mass-flow-rate BCs with window data: BCOutMFR on imax (qp per face cell), BCInjMFR on imin (direction, qp and
    stagnation enthalpy per face cell [, nu_tilde]) and, with rest_face, on kmin, where the Taylor-Green field has no
    normal velocity, so that the first ghost fill takes the Newton start of BCInjMFR.for."""
    ni, nj, nk = n
    neq = 6 if model == 'NSTurbulent' else 5
    cv = 1. / (1.4 * 0.4 * 1.4 * 0.01)
    hinf = 1.4 * cv * 1. + 0.5 * 1.
    bcs = {'imax': ('BCOutMFR', {'qp': window_data((nj * nk,), 0.6, 0.05)}), 'jmin': 'BCWall', 'jmax': 'BCFarfield'}
    def inj(ng, d, q):
        dd = {'d0x': numpy_full(ng, d[0]), 'd0y': numpy_full(ng, d[1]), 'd0z': window_data((ng,), d[2], 0.1),
              'qp': window_data((ng,), q, 0.03), 'StagnationEnthalpy': window_data((ng,), hinf, 0.001)}
        if neq == 6:
            dd['TurbulentSANuTilde'] = window_data((ng,), 2e-5, 0.3)
        return dd
    bcs['imin'] = ('BCInjMFR', inj(nj * nk, (0.99, 0.05, 0.02), 0.7))
    if rest_face:
        bcs['kmin'] = ('BCInjMFR', inj(ni * nj, (0.05, 0.02, 0.99), 0.3))
    return bcs
