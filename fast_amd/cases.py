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
        perturb=0., extra_numb=None):
    """Single- or multi-block TGV tree (+ numerics).  n = interior cells per direction of the
    whole box; blocks = (bi,bj,bk) split; only = block numbers to instantiate (the blocks of one
    rank; block b = ib + jb*bi + kb*bi*bj is named 'cart<b>').  Split directions carry no BC:
    their ghost cells are filled by match joins (fast_amd.connect).  perturb > 0 displaces the nodes
    smoothly (see _perturb) so that the zones are curvilinear.  Returns t."""
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
                    _perturb(z, perturb, L1)
                faces = []
                if blocks[0] == 1: faces += ['imin', 'imax']
                if blocks[1] == 1: faces += ['jmin', 'jmax']
                if blocks[2] == 1: faces += ['kmin', 'kmax']
                if bcs:
                    for f in [f for f in faces if f in bcs]:
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
        sa_fields(t, state, [f for f, b in (bcs or {}).items() if b.startswith('BCWall')], L1)
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
          extra_numz=None, extra_numb=None, bcs=None, perturb=0.):
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
