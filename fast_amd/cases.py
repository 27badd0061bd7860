"""Synthetic FastS cases (deterministic, no RNG) in the pyTree layout of the reference's
own scripts.

* ``tgv``    : Taylor-Green vortex on [-pi,pi]^3, field formulas of
               Fast/FastS/test/tgvMpi.py:54-58, NSLaminar, M=0.1, Re=1600, 6 x BCautoperiod
               (tgvMpi.py:66-74) or split in match-joined blocks.
* ``vortex`` : inviscid isentropic vortex advected along x on a periodic box
               (the 64^3 Euler / Roe / RK3 case).
Zones carry their two ghost layers in the mesh, as in tgvMpi.py:41-50.
"""
import math
import numpy
from . import Internal
from . import pytree_lite as C
from . import FastC


def _box(n, length, origin, name):
    """n interior cells per direction + 2 ghost layers each side."""
    h = [length[d] / n[d] for d in range(3)]
    o = [origin[d] - 2 * h[d] for d in range(3)]
    return C.cart(o, h, (n[0] + 5, n[1] + 5, n[2] + 5), name=name)


def _perturb(z, amp, L=2. * math.pi):
    """Smooth node displacement of period L in every direction (amp in units of the local mesh
    step): the zone stops being Cartesian and is classified curvilinear (zone type 0), while the
    ghost nodes stay periodic images of interior nodes up to round-off."""
    g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
    xs = [Internal.getNodeFromName1(g, n)[1] for n in ('CoordinateX', 'CoordinateY', 'CoordinateZ')]
    h = [xs[0][1, 0, 0] - xs[0][0, 0, 0], xs[1][0, 1, 0] - xs[1][0, 0, 0], xs[2][0, 0, 1] - xs[2][0, 0, 0]]
    w = 2. * math.pi / L
    x, y, zc = [a.copy() for a in xs]
    xs[0][...] = x + amp * h[0] * numpy.sin(w * y + 0.3) * numpy.cos(w * zc) * numpy.cos(w * x)
    xs[1][...] = y + amp * h[1] * numpy.sin(w * zc + 0.7) * numpy.cos(w * x) * numpy.cos(w * y)
    xs[2][...] = zc + amp * h[2] * numpy.sin(w * x + 1.1) * numpy.cos(w * y) * numpy.cos(w * zc)


def _perturb_homo(z, amp, L=2. * math.pi):
    """Smooth displacement of the nodes inside the planes z = const, the same in every plane: an extruded mesh,
    classified k-homogeneous (zone type 1: metrics of one plane)."""
    g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
    xs = [Internal.getNodeFromName1(g, n)[1] for n in ('CoordinateX', 'CoordinateY')]
    h = [xs[0][1, 0, 0] - xs[0][0, 0, 0], xs[1][0, 1, 0] - xs[1][0, 0, 0]]
    w = 2. * math.pi / L
    x, y = xs[0][:, :, 0].copy(), xs[1][:, :, 0].copy()
    xs[0][...] = (x + amp * h[0] * numpy.sin(w * y + 0.3) * numpy.cos(w * x))[:, :, None]
    xs[1][...] = (y + amp * h[1] * numpy.sin(w * x + 0.7) * numpy.cos(w * y))[:, :, None]


def _collapse(z, faces, ghost=2):
    """Put the ghost nodes of the given faces on the boundary nodes: the flat ghost cells a mesh generator leaves
    behind physical boundaries (the solver then takes their metrics from the mirror cells)."""
    g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
    for name in ('CoordinateX', 'CoordinateY', 'CoordinateZ'):
        a = Internal.getNodeFromName1(g, name)[1]
        for f in faces:
            d = 'ijk'.index(f[0]); n = a.shape[d]
            for r in range(ghost):
                src = [slice(None)] * 3; dst = [slice(None)] * 3
                src[d] = ghost if f.endswith('min') else n - 1 - ghost
                dst[d] = r if f.endswith('min') else n - 1 - r
                a[tuple(dst)] = a[tuple(src)]


def _periodic_faces(z, faces=('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')):
    for f in faces:
        C.addBC2Zone(z, 'period ' + f, 'BCautoperiod', f)


def tgv_fields(t, state, L0=1.):
    """Density / velocity / temperature of tgvMpi.py:54-58 with P0, T0 taken from the
    reference state (rho0 = U0 = T0 = 1)."""
    P0 = state[5]; RO0 = state[0]; T0 = state[6]; U0 = 1.
    K0 = RO0 * U0 * U0 / 16.
    O0 = P0 / (RO0 * T0) * T0
    C._initVars(t, 'centers:Density',
                lambda x, y, z: (P0 + K0 * (numpy.cos(2. * x / L0) + numpy.cos(2. * y / L0))
                                 * (numpy.cos(2. * z / L0) + 2.)) / O0)
    C._initVars(t, 'centers:VelocityX',
                lambda x, y, z: U0 * numpy.sin(x / L0) * numpy.cos(y / L0) * numpy.cos(z / L0))
    C._initVars(t, 'centers:VelocityY',
                lambda x, y, z: -U0 * numpy.cos(x / L0) * numpy.sin(y / L0) * numpy.cos(z / L0))
    C._initVars(t, 'centers:VelocityZ', 0.)
    C._initVars(t, 'centers:Temperature', T0)


def sa_fields(t, state, walls=(), L=2. * math.pi):
    """Spalart-Allmaras inputs of a box test: a smooth positive nu_tilde of a few laminar viscosities and
    the wall distance (distance to the nearest of the faces in `walls`, or a smooth positive field when the
    box has no wall: only the value matters to the source term)."""
    nu = state[15] / state[0]          # laminar kinematic viscosity of the reference state (Mus / rho)
    C._initVars(t, 'centers:TurbulentSANuTilde',
                lambda x, y, z: 3. * nu * (1.5 + numpy.sin(x) * numpy.cos(y) + 0.3 * numpy.cos(2. * z)))
    h = 0.5 * L

    def dist(x, y, z):
        d = None
        for f in walls:
            c = {'i': x, 'j': y, 'k': z}[f[0]]
            df = (c + h) if f.endswith('min') else (h - c)
            d = df if d is None else numpy.minimum(d, df)
        if d is None:
            d = 1. + 0.3 * numpy.sin(x) * numpy.sin(y)
        return numpy.abs(d) + 0. * x
    C._initVars(t, 'centers:TurbulentDistance', dist)


def tgv(n=32, scheme='senseur', model='NSLaminar', time_step=None, blocks=(1, 1, 1),
        temporal_scheme='explicit', modulo_verif=50, extra_numz=None, only=None, bcs=None,
        perturb=0., extra_numb=None, collapse=(), homo=False):
    """Single- or multi-block TGV tree (+ numerics).  n = interior cells per direction of the
    whole box; blocks = (bi,bj,bk) split; only = block numbers to instantiate (the blocks of one
    rank; block b = ib + jb*bi + kb*bi*bj is named 'cart<b>').  Split directions carry no BC:
    their ghost cells are filled by match joins (fast_amd.connect).  perturb > 0 displaces the nodes
    smoothly (see _perturb) so that the zones are curvilinear; with homo the displacement is that of an extruded
    mesh (_perturb_homo, zone type 1).  Returns t."""
    L1 = 2. * math.pi
    nn = (n, n, n) if isinstance(n, int) else tuple(n)
    nb = [nn[d] // blocks[d] for d in range(3)]
    n = max(nn)
    zones = []
    for kb in range(blocks[2]):
        for jb in range(blocks[1]):
            for ib in range(blocks[0]):
                org = [-0.5 * L1 + ib * L1 / blocks[0], -0.5 * L1 + jb * L1 / blocks[1],
                       -0.5 * L1 + kb * L1 / blocks[2]]
                ln = [L1 / blocks[0], L1 / blocks[1], L1 / blocks[2]]
                b = ib + jb * blocks[0] + kb * blocks[0] * blocks[1]
                if only is not None and b not in only:
                    continue
                name = 'cart%d' % b
                z = _box(nb, ln, org, name)
                if perturb:
                    (_perturb_homo if homo else _perturb)(z, perturb, L1)
                if collapse:
                    _collapse(z, collapse)
                faces = []
                if blocks[0] == 1: faces += ['imin', 'imax']
                if blocks[1] == 1: faces += ['jmin', 'jmax']
                if blocks[2] == 1: faces += ['kmin', 'kmax']
                if bcs:
                    for f in [f for f in faces if f in bcs]:
                        if isinstance(bcs[f], tuple):    # (type, {name: window data})
                            C.addBC2Zone(z, bcs[f][0].lower() + ' ' + f, bcs[f][0], f, data=bcs[f][1])
                        else:
                            C.addBC2Zone(z, bcs[f].lower() + ' ' + f, bcs[f], f)
                    faces = [f for f in faces if f not in bcs]
                _periodic_faces(z, faces)
                zones.append(z)
    t = C.newPyTree(['Base'] + zones)
    C.addState(t, 'GoverningEquations', model)
    C.addState(t, adim='adim2funk', MInf=0.1, ReInf=1600.)
    state = C.getState(Internal.getNodeFromName(t, 'ReferenceState'))
    tgv_fields(t, state)
    if model in ('NSTurbulent', 'nsspalart'):
        sa_fields(t, state, [f for f, b in (bcs or {}).items()
                             if (b[0] if isinstance(b, tuple) else b).startswith('BCWall')], L1)
    if time_step is None:
        time_step = 0.003 * 32. / n          # tgvMpi.py:271 scaled with the mesh size
    numb = {'temporal_scheme': temporal_scheme, 'modulo_verif': modulo_verif}
    numz = {'time_step': time_step, 'scheme': scheme}
    if extra_numz:
        numz.update(extra_numz)
    if extra_numb:
        numb.update(extra_numb)
    FastC._setNum2Zones(t, numz)
    FastC._setNum2Base(t, numb)
    return t


def tgv2d(n=(16, 12), scheme='ausmpred', model='NSLaminar', time_step=None, temporal_scheme='explicit',
          extra_numz=None, extra_numb=None, bcs=None, perturb=0., collapse=()):
    """Two-dimensional zone (one cell along k, no ghost layer along k: zone type 3) on [-pi,pi]^2 with the
    Taylor-Green fields of the plane z = 0; faces without an entry in bcs are periodic.  perturb displaces the
    nodes in the plane."""
    L1 = 2. * math.pi
    h = [L1 / n[0], L1 / n[1]]
    z = C.cart((-0.5 * L1 - 2 * h[0], -0.5 * L1 - 2 * h[1], 0.), (h[0], h[1], min(h)), (n[0] + 5, n[1] + 5, 2), name='plane0')
    if perturb:
        g = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
        xs = [Internal.getNodeFromName1(g, c)[1] for c in ('CoordinateX', 'CoordinateY')]
        x, y = xs[0].copy(), xs[1].copy()
        xs[0][...] = x + perturb * h[0] * numpy.sin(y + 0.3) * numpy.cos(x)
        xs[1][...] = y + perturb * h[1] * numpy.sin(x + 0.7) * numpy.cos(y)
    if collapse:
        _collapse(z, collapse)
    faces = ['imin', 'imax', 'jmin', 'jmax']
    if bcs:
        for f in [f for f in faces if f in bcs]:
            C.addBC2Zone(z, bcs[f].lower() + ' ' + f, bcs[f], f)
        faces = [f for f in faces if f not in bcs]
    _periodic_faces(z, faces)
    t = C.newPyTree(['Base', z])
    C.addState(t, 'GoverningEquations', model)
    C.addState(t, adim='adim2funk', MInf=0.1, ReInf=1600.)
    state = C.getState(Internal.getNodeFromName(t, 'ReferenceState'))
    tgv_fields(t, state)
    C._initVars(t, 'centers:VelocityZ', 0.)
    if model in ('NSTurbulent', 'nsspalart'):
        sa_fields(t, state, [f for f, b in (bcs or {}).items() if b.startswith('BCWall')], L1)
    if time_step is None:
        time_step = 0.003 * 32. / max(n)
    numb = {'temporal_scheme': temporal_scheme, 'modulo_verif': 50}
    numz = {'time_step': time_step, 'scheme': scheme}
    if extra_numz:
        numz.update(extra_numz)
    if extra_numb:
        numb.update(extra_numb)
    FastC._setNum2Zones(t, numz)
    FastC._setNum2Base(t, numb)
    return t


def _naca0012_y(x):
    """Half thickness of the NACA0012 section with a closed trailing edge, chord 1."""
    x = numpy.clip(x, 0., 1.)
    return 0.6 * (0.2969 * numpy.sqrt(x) - 0.126 * x - 0.3516 * x ** 2 + 0.2843 * x ** 3 - 0.1036 * x ** 4)


def naca_cmesh(nfoil=32, nwake=12, nj=16, radius=8., wall_step=2.e-3, scheme='ausmpred', model='NSTurbulent',
               temporal_scheme='implicit', ss_iteration=3, cfl=5., mach=0.3, reynolds=1.e5, extra_numz=None,
               extra_numb=None):
    """Two-dimensional C-mesh around a NACA0012 section, built analytically (no mesh file): i runs from the
    downstream end of the lower wake line, around the section (lower side, leading edge, upper side) to the
    downstream end of the upper wake line; j from the section / wake cut to a C-shaped outer boundary of the
    given radius centred on the trailing edge.  2*nwake + 2*nfoil cells along i, nj along j, one along k
    (zone type 3).  jmin carries a wall on the section and a match join between the two sides of the wake cut
    (see connect.cmesh_wake_join); the other faces are far field.  Steady-RANS set-up of the reference
    (implicit, local time step, Spalart-Allmaras) by default.  Returns t."""
    ni = 2 * nwake + 2 * nfoil
    # inner line: nodes 0..ni
    # section abscissa TE -> LE, clustered at the leading edge only (a closed trailing edge is too thin for
    # clustered cells: their ghost cells would fold); wake abscissa stretched from the trailing-edge cell size
    xf = 1. - numpy.sin(0.5 * math.pi * numpy.linspace(0., 1., nfoil + 1))
    xf[-1] = 0.
    pw = math.log((1. - xf[1]) / (radius - 1.)) / math.log(1. / nwake)
    xw = 1. + (radius - 1.) * (numpy.linspace(0., 1., nwake + 1) ** pw)
    xin = numpy.concatenate([xw[::-1], xf[1:], xf[::-1][1:], xw[1:]])
    yin = numpy.concatenate([numpy.zeros(nwake + 1), -_naca0012_y(xf[1:]), _naca0012_y(xf[::-1][1:]), numpy.zeros(nwake)])
    yin[nwake + nfoil] = 0.
    # outer line: straight above/below the wake, half circle of centre (1, 0) around the section
    ang = numpy.linspace(-0.5 * math.pi, -1.5 * math.pi, 2 * nfoil + 1)
    xout = numpy.concatenate([xw[::-1], 1. + radius * numpy.cos(ang[1:]), xw[1:]])
    yout = numpy.concatenate([-radius * numpy.ones(nwake + 1), radius * numpy.sin(ang[1:]), radius * numpy.ones(nwake)])
    # geometric stretching along j
    q = 1.
    for _ in range(200):        # growth rate q with first step wall_step and total length 1 (in units of radius)
        tot = wall_step * (nj if abs(q - 1.) < 1e-12 else (q ** nj - 1.) / (q - 1.))
        q = q * (1. / tot) ** (1. / nj) if tot > 0 else q
        if abs(tot - 1.) < 1e-12:
            break
    sj = numpy.concatenate([[0.], numpy.cumsum(wall_step * q ** numpy.arange(nj))])
    sj = sj / sj[-1]
    g = 2
    X = numpy.zeros((ni + 1 + 2 * g, nj + 1 + 2 * g, 2)); Y = numpy.zeros_like(X); Z = numpy.zeros_like(X)
    for jj in range(nj + 1):
        X[g:g + ni + 1, g + jj, :] = (xin + sj[jj] * (xout - xin))[:, None]
        Y[g:g + ni + 1, g + jj, :] = (yin + sj[jj] * (yout - yin))[:, None]
    for A in (X, Y):            # ghost nodes: linear extrapolation, j then i
        for r in range(1, g + 1):
            A[:, g - r, :] = 2. * A[:, g - r + 1, :] - A[:, g - r + 2, :]
            A[:, g + nj + r, :] = 2. * A[:, g + nj + r - 1, :] - A[:, g + nj + r - 2, :]
        for r in range(1, g + 1):
            A[g - r, :, :] = 2. * A[g - r + 1, :, :] - A[g - r + 2, :, :]
            A[g + ni + r, :, :] = 2. * A[g + ni + r - 1, :, :] - A[g + ni + r - 2, :, :]
    for r in range(1, g + 1):   # across the wake cut the ghost nodes are the nodes of the other side
        for i in list(range(0, nwake + 1)) + list(range(ni - nwake, ni + 1)):
            X[g + i, g - r, :] = X[g + ni - i, g + r, :]
            Y[g + i, g - r, :] = Y[g + ni - i, g + r, :]
    hz = 1.
    Z[:, :, 1] = hz
    z = C.cart((0., 0., 0.), (1., 1., 1.), (ni + 1 + 2 * g, nj + 1 + 2 * g, 2), name='cmesh')
    gc = Internal.getNodeFromName1(z, Internal.__GridCoordinates__)
    for name, A in (('CoordinateX', X), ('CoordinateY', Y), ('CoordinateZ', Z)):
        Internal.getNodeFromName1(gc, name)[1][...] = A
    a, b = nwake + 1, ni - nwake           # section cells along i (1-based interior numbering)
    C.addBC2Zone(z, 'wall', 'BCWall', [a + g, b + g + 1, 1, 1, 1, 2])
    C.addBC2Zone(z, 'far jmax', 'BCFarfield', 'jmax')
    C.addBC2Zone(z, 'far imin', 'BCFarfield', 'imin')
    C.addBC2Zone(z, 'far imax', 'BCFarfield', 'imax')
    t = C.newPyTree(['Base', z])
    C.addState(t, 'GoverningEquations', model)
    C.addState(t, adim='adim1', MInf=mach, ReInf=reynolds)
    state = C.getState(Internal.getNodeFromName(t, 'ReferenceState'))
    C._initVars(t, 'centers:Density', state[0])
    C._initVars(t, 'centers:VelocityX', state[1] / state[0])
    C._initVars(t, 'centers:VelocityY', 0.)
    C._initVars(t, 'centers:VelocityZ', 0.)
    C._initVars(t, 'centers:Temperature', state[6])
    if model in ('NSTurbulent', 'nsspalart'):
        nu = state[15] / state[0]
        C._initVars(t, 'centers:TurbulentSANuTilde', 3. * nu)
        # wall distance: to the section for the columns that start on it, to the trailing edge in the wake
        xc = 0.25 * (X[:-1, :-1, 0] + X[1:, :-1, 0] + X[:-1, 1:, 0] + X[1:, 1:, 0])
        yc = 0.25 * (Y[:-1, :-1, 0] + Y[1:, :-1, 0] + Y[:-1, 1:, 0] + Y[1:, 1:, 0])
        xb = 0.5 * (X[:-1, g, 0] + X[1:, g, 0]); yb = 0.5 * (Y[:-1, g, 0] + Y[1:, g, 0])
        d = numpy.sqrt((xc - xb[:, None]) ** 2 + (yc - yb[:, None]) ** 2)
        dte = numpy.sqrt((xc - 1.) ** 2 + yc ** 2)
        wake = numpy.ones(ni + 2 * g, dtype=bool); wake[g + a - 1:g + b] = False
        d[wake, :] = dte[wake, :]
        sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
        C._initVars(t, 'centers:TurbulentDistance', 0.)
        Internal.getNodeFromName1(sol, 'TurbulentDistance')[1][:, :, 0] = numpy.maximum(d, 1.e-8)
    numb = {'temporal_scheme': temporal_scheme, 'modulo_verif': 50, 'ss_iteration': ss_iteration}
    numz = {'time_step': 1.e-3, 'scheme': scheme}
    if temporal_scheme == 'implicit':
        numz.update({'time_step_nature': 'local', 'cfl': cfl})
    if extra_numz:
        numz.update(extra_numz)
    if extra_numb:
        numb.update(extra_numb)
    FastC._setNum2Zones(t, numz)
    FastC._setNum2Base(t, numb)
    return t


def vortex(n=64, scheme='roe', time_step=None, extra_numz=None, extra_numb=None):
    """Isentropic vortex (strength beta=5, core radius 1) in a 10^3 periodic box, free
    stream along x at M=0.5, Euler, sound-speed scaling (rho_inf = c_inf = T_inf = 1)."""
    Lb = 10.
    z = _box((n, n, n), (Lb, Lb, Lb), (0., 0., 0.), 'cart')
    _periodic_faces(z)
    t = C.newPyTree(['Base', z])
    C.addState(t, 'GoverningEquations', 'Euler')
    C.addState(t, adim='adim1', MInf=0.5, ReInf=1.e8)
    st = C.getState(Internal.getNodeFromName(t, 'ReferenceState'))
    gam = st[11]; cv = st[7]; rgp = cv * (gam - 1.)
    roinf, pinf, Tinf = st[0], st[5], st[6]
    uinf = st[1] / st[0]
    beta = 5.; xc = yc = 0.5 * Lb

    def dT(x, y):
        r2 = (x - xc) ** 2 + (y - yc) ** 2
        return -(gam - 1.) * beta * beta / (8. * gam * math.pi ** 2) * numpy.exp(1. - r2)

    def T(x, y, z):
        return Tinf * (1. + dT(x, y) * gam)   # T_inf = 1, c_inf^2 = gamma R T_inf = 1

    def ro(x, y, z):
        return roinf * (T(x, y, z) / Tinf) ** (1. / (gam - 1.))

    def u(x, y, z):
        r2 = (x - xc) ** 2 + (y - yc) ** 2
        return uinf - beta / (2. * math.pi) * numpy.exp(0.5 * (1. - r2)) * (y - yc)

    def v(x, y, z):
        r2 = (x - xc) ** 2 + (y - yc) ** 2
        return beta / (2. * math.pi) * numpy.exp(0.5 * (1. - r2)) * (x - xc)

    C._initVars(t, 'centers:Density', ro)
    C._initVars(t, 'centers:VelocityX', u)
    C._initVars(t, 'centers:VelocityY', v)
    C._initVars(t, 'centers:VelocityZ', 0.)
    C._initVars(t, 'centers:Temperature', T)
    if time_step is None:
        time_step = 0.5 * (Lb / n) / (uinf + 1.)     # CFL ~ 0.5 on (u+c)
    numb = {'temporal_scheme': 'explicit', 'modulo_verif': 50}
    numz = {'time_step': time_step, 'scheme': scheme}
    if extra_numz:
        numz.update(extra_numz)
    if extra_numb:
        numb.update(extra_numb)
    FastC._setNum2Zones(t, numz)
    FastC._setNum2Base(t, numb)
    return t
