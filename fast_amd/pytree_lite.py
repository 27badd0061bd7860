"""Stand-alone subset of the Cassiopee pre-processing calls the reference test scripts use
to *build* a FastS case: Generator.cart, Converter.newPyTree/addState/addBC2Zone/
_initVars/node2Center and Initiator.Adim states (call sites: Fast/FastS/test/tgvMpi.py:50-78).

Only the data layout matters to the solver; these helpers produce the same CGNS/Python
node layout (names, shapes, Fortran ordering) so scripts written against Cassiopee can be
fed to fast_amd without it.  Cassiopee itself is not part of the reference tree, so the
numerical values of its reference-state helpers are *defined here*, not reproduced.
"""
import math
import numpy
from . import Internal

_STATE_NAMES = ['Density', 'MomentumX', 'MomentumY', 'MomentumZ',
                'EnergyStagnationDensity', 'Pressure', 'Temperature', 'Cv', 'Mach',
                'Reynolds', 'Cs_', 'Gamma', 'Rok', 'RoOmega',
                'TurbulentSANuTildeDensity', 'Mus', 'Cs', 'Ts', 'Pr']


def cart(origin, h, n, name='cart'):
    """Structured Cartesian zone with n=(ni,nj,nk) nodes (Generator.cart semantics)."""
    ni, nj, nk = int(n[0]), int(n[1]), int(n[2])
    dims = numpy.array([[ni, ni - 1, 0], [nj, nj - 1, 0], [nk, nk - 1, 0]],
                       dtype=Internal.E_NpyInt, order='F')
    z = Internal.createNode(name, 'Zone_t', dims)
    Internal.createChild(z, 'ZoneType', 'ZoneType_t', 'Structured')
    g = Internal.createChild(z, Internal.__GridCoordinates__, 'GridCoordinates_t')
    i = numpy.arange(ni, dtype=numpy.float64)
    j = numpy.arange(nj, dtype=numpy.float64)
    k = numpy.arange(nk, dtype=numpy.float64)
    x = numpy.empty((ni, nj, nk), dtype=numpy.float64, order='F')
    y = numpy.empty((ni, nj, nk), dtype=numpy.float64, order='F')
    zc = numpy.empty((ni, nj, nk), dtype=numpy.float64, order='F')
    x[:, :, :] = (origin[0] + i * h[0])[:, None, None]
    y[:, :, :] = (origin[1] + j * h[1])[None, :, None]
    zc[:, :, :] = (origin[2] + k * h[2])[None, None, :]
    Internal.createChild(g, 'CoordinateX', 'DataArray_t', x)
    Internal.createChild(g, 'CoordinateY', 'DataArray_t', y)
    Internal.createChild(g, 'CoordinateZ', 'DataArray_t', zc)
    return z


def newPyTree(args):
    """newPyTree(['Base', z1, z2, ...]) or newPyTree(['Base'])."""
    t = Internal.newCGNSTree()
    base = None
    for a in args:
        if isinstance(a, str):
            base = Internal.createChild(t, a, 'CGNSBase_t',
                                        numpy.array([3, 3], dtype=Internal.E_NpyInt))
        elif isinstance(a, list):
            if len(a) == 4 and isinstance(a[0], str):
                base[2].append(a)
            else:
                base[2] += a
    return t


def center_coordinates(z):
    """Cell-centre coordinates (ni-1,nj-1,nk-1), Fortran order."""
    g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
    out = []
    for n in ('CoordinateX', 'CoordinateY', 'CoordinateZ'):
        a = Internal.getNodeFromName1(g, n)[1]
        c = 0.125 * (a[:-1, :-1, :-1] + a[1:, :-1, :-1] + a[:-1, 1:, :-1] + a[1:, 1:, :-1]
                     + a[:-1, :-1, 1:] + a[1:, :-1, 1:] + a[:-1, 1:, 1:] + a[1:, 1:, 1:])
        out.append(numpy.asfortranarray(c))
    return out


def _initVars(t, name, value):
    """Set a field: name is 'centers:Var' (or 'Var' for nodes); value is a constant or a
    callable f(x, y, z) evaluated at the cell centres (resp. nodes)."""
    loc, var = name.split(':') if ':' in name else ('nodes', name)
    for z in Internal.getZones(t):
        d = Internal.getZoneDim(z)
        if loc == 'centers':
            cont = Internal.getNodeFromName1(z, Internal.__FlowSolutionCenters__)
            if cont is None:
                cont = Internal.createChild(z, Internal.__FlowSolutionCenters__, 'FlowSolution_t')
                Internal.createChild(cont, 'GridLocation', 'GridLocation_t', 'CellCenter')
            shape = (d[1] - 1, d[2] - 1, d[3] - 1)
            if callable(value):
                xc, yc, zc = center_coordinates(z)
                a = numpy.asfortranarray(value(xc, yc, zc), dtype=numpy.float64)
                if a.shape != shape:
                    a = numpy.asfortranarray(numpy.broadcast_to(a, shape).copy())
            else:
                a = numpy.full(shape, float(value), dtype=numpy.float64, order='F')
        else:
            cont = Internal.getNodeFromName1(z, Internal.__FlowSolutionNodes__)
            if cont is None:
                cont = Internal.createChild(z, Internal.__FlowSolutionNodes__, 'FlowSolution_t')
            shape = (d[1], d[2], d[3])
            g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
            if callable(value):
                xs = [Internal.getNodeFromName1(g, n)[1]
                      for n in ('CoordinateX', 'CoordinateY', 'CoordinateZ')]
                a = numpy.asfortranarray(value(*xs), dtype=numpy.float64)
            else:
                a = numpy.full(shape, float(value), dtype=numpy.float64, order='F')
        Internal.createUniqueChild(cont, var, 'DataArray_t', a)
    return None


def _state_node(parent):
    ref = Internal.getNodeFromName1(parent, 'ReferenceState')
    if ref is None:
        ref = Internal.createChild(parent, 'ReferenceState', 'ReferenceState_t')
    return ref


def addState(t, state=None, value=None, adim=None, **kw):
    """addState(t, 'GoverningEquations', 'NSLaminar') or addState(t, adim='adim2funk',
    MInf=..., ReInf=...) or addState(t, state=<19-vector>)."""
    bases = Internal.getBases(t)
    if isinstance(state, str):
        for b in bases:
            if state == 'GoverningEquations':
                fe = Internal.getNodeFromName1(b, 'FlowEquationSet')
                if fe is None:
                    fe = Internal.createChild(b, 'FlowEquationSet', 'FlowEquationSet_t')
                Internal.createUniqueChild(fe, 'GoverningEquations', 'GoverningEquations_t', value)
            elif state == 'EquationDimension':
                fe = Internal.getNodeFromName1(b, 'FlowEquationSet')
                if fe is None:
                    fe = Internal.createChild(b, 'FlowEquationSet', 'FlowEquationSet_t')
                Internal.createUniqueChild(fe, 'EquationDimension', '"int"', value)
            else:
                Internal.createUniqueChild(_state_node(b), state, 'DataArray_t', value)
        return t
    if adim is not None:
        fn = {'adim1': adim1, 'adim2': adim2funk, 'adim2funk': adim2funk}[adim]
        state = fn(**kw)
    if state is not None:
        for b in bases:
            ref = _state_node(b)
            for n, v in zip(_STATE_NAMES, state):
                Internal.createUniqueChild(ref, n, 'DataArray_t', float(v))
    return t


def getState(ref):
    """19-entry reference state vector from a ReferenceState node: RoInf, RouInf, RovInf,
    RowInf, RoEInf, PInf, TInf, cvInf, MInf, ReInf, Cs, Gamma, RokInf, RoomegaInf,
    RonutildeInf, Mus, Cs, Ts, Pr (order documented at
    Fast/FastS/test/explicitLocalTgv_t1.py:78 and consumed at
    Fast/FastC/FastC/PyTree.py:1956-1975)."""
    if ref[3] != 'ReferenceState_t':
        ref = Internal.getNodeFromName(ref, 'ReferenceState')
    out = []
    for n in _STATE_NAMES:
        c = Internal.getNodeFromName1(ref, n)
        out.append(float(Internal.getValue(c)) if c is not None else 0.)
    return out


def _pack_state(ro, u, v, w, p, T, cv, M, Re, gamma, mus, Cs, Ts, Pr, ronut=None):
    roE = p / (gamma - 1.) + 0.5 * ro * (u * u + v * v + w * w)
    if ronut is None:
        ronut = 0.1 * mus  # rho*nutilde_inf ~ 0.1 mu_inf: usual free-stream SA level
    return [ro, ro * u, ro * v, ro * w, roE, p, T, cv, M, Re, Cs, gamma, 0., 0., ronut,
            mus, Cs, Ts, Pr]


def adim2funk(MInf=0.5, alphaZ=0., alphaY=0., ReInf=1.e8, **_):
    """Reference state scaled by density, flow speed and temperature
    (rho_inf = U_inf = T_inf = 1, L = 1): p_inf = 1/(gamma M^2), mu_inf = 1/Re.
    Sutherland law anchored at T_inf (Ts = 1, Cs = 110.4 K / 298.15 K)."""
    gamma = 1.4
    ro, U, T = 1., 1., 1.
    az, ay = math.radians(alphaZ), math.radians(alphaY)
    u = U * math.cos(az) * math.cos(ay)
    v = U * math.sin(az) * math.cos(ay)
    w = U * math.sin(ay)
    p = ro * U * U / (gamma * MInf * MInf)
    cv = p / (ro * T * (gamma - 1.))
    mus = ro * U / ReInf
    return _pack_state(ro, u, v, w, p, T, cv, MInf, ReInf, gamma, mus, 110.4 / 298.15, 1., 0.72)


def adim1(MInf=0.5, alphaZ=0., alphaY=0., ReInf=1.e8, **_):
    """Reference state scaled by density, sound speed and temperature
    (rho_inf = c_inf = T_inf = 1): p_inf = 1/gamma, U_inf = M."""
    gamma = 1.4
    ro, T = 1., 1.
    U = MInf
    az, ay = math.radians(alphaZ), math.radians(alphaY)
    u = U * math.cos(az) * math.cos(ay)
    v = U * math.sin(az) * math.cos(ay)
    w = U * math.sin(ay)
    p = 1. / gamma
    cv = p / (ro * T * (gamma - 1.))
    mus = ro * max(U, 1e-30) / ReInf
    return _pack_state(ro, u, v, w, p, T, cv, MInf, ReInf, gamma, mus, 110.4 / 298.15, 1., 0.72)


_WINDOWS = {'imin': 0, 'imax': 1, 'jmin': 2, 'jmax': 3, 'kmin': 4, 'kmax': 5}


def addBC2Zone(z, bndName, bndType, wrange, data=None):
    """Add a BC_t on a full face ('imin'...'kmax') or an explicit node window
    [i1,i2,j1,j2,k1,k2] (1-based node indices, Converter.addBC2Zone semantics).  data = {name: array}
    attaches window data (one value per face cell of the window, first transverse direction fastest) as
    BCDataSet_t / BCData_t / DataArray_t, the nesting the solver's BC table is built from."""
    d = Internal.getZoneDim(z)
    ni, nj, nk = d[1], d[2], d[3]
    if isinstance(wrange, str):
        w = [1, ni, 1, nj, 1, nk]
        f = _WINDOWS[wrange]
        if f == 0: w[1] = 1
        elif f == 1: w[0] = ni
        elif f == 2: w[3] = 1
        elif f == 3: w[2] = nj
        elif f == 4: w[5] = 1
        else: w[4] = nk
    else:
        w = list(wrange)
    zbc = Internal.getNodeFromName1(z, 'ZoneBC')
    if zbc is None:
        zbc = Internal.createChild(z, 'ZoneBC', 'ZoneBC_t')
    bc = Internal.createChild(zbc, bndName, 'BC_t', bndType)
    pr = numpy.array([[w[0], w[1]], [w[2], w[3]], [w[4], w[5]]], dtype=Internal.E_NpyInt, order='F')
    Internal.createChild(bc, 'PointRange', 'IndexRange_t', pr)
    if data:
        ds = Internal.createChild(bc, 'BCDataSet', 'BCDataSet_t', bndType)
        bd = Internal.createChild(ds, 'DirichletData', 'BCData_t', None)
        for name, val in data.items():
            Internal.createChild(bd, name, 'DataArray_t', numpy.ascontiguousarray(val, dtype=numpy.float64).ravel())
    return z


def node2Center(t):
    """Connectivity-tree skeleton: same bases/zones, zone dims turned into cell counts,
    no field containers (what the solver needs from Converter.node2Center + rmVars)."""
    tc = Internal.newCGNSTree()
    for b in Internal.getBases(t):
        bc = Internal.createChild(tc, b[0], 'CGNSBase_t', b[1])
        for z in Internal.getZones(b):
            d = z[1]
            dims = numpy.array([[d[0][1], max(d[0][1] - 1, 0), 0],
                                [d[1][1], max(d[1][1] - 1, 0), 0],
                                [d[2][1], max(d[2][1] - 1, 0), 0]],
                               dtype=Internal.E_NpyInt, order='F')
            zc = Internal.createChild(bc, z[0], 'Zone_t', dims)
            Internal.createChild(zc, 'ZoneType', 'ZoneType_t', 'Structured')
    return tc
