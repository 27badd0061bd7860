"""Host-side solver state for the FastS time step: the mirror of the reference's
``FastC.PyTree`` services that ``FastS.PyTree.warmup`` relies on.

What is restated here, with the reference location each part follows:

* numerics dictionaries -> ``.Solver#define`` nodes        Fast/FastC/FastC/PyTree.py:107-135
* ``_buildOwnData``: ``.Solver#define`` + ReferenceState -> ``Parameter_int`` /
  ``Parameter_real`` / ``.Solver#dtloc``                   Fast/FastC/FastC/PyTree.py:918-2053
* zone sizes, zone type, metric sizes (what ``fasts.allocate_metric`` writes into
  ``Parameter_int``)                Fast/FastS/FastS/Metric/allocate_metric.cpp:124-256,398-462
  and Fast/FastC/FastC/Metric/nature_geom_dom.for:186-503
* metrics of a Cartesian zone       Fast/FastC/FastC/Metric/cp_tijk.for:291-322, cp_vol.for:148-175
* BC table appended to ``Parameter_int``   Fast/FastC/FastC/PyTree.py:2789-3017 and
  Fast/FastC/FastC/getRange.cpp:57-115
* primitive variables / three time levels / compaction     Fast/FastC/FastC/PyTree.py:214-640,2230-2271
* time-level pointer rotation                               Fast/FastC/FastC/PyTree.py:2292-2315

The arrays are laid out exactly as the reference lays them out (index constants below are
those of Fast/FastS/FastS/param_solver.h:77-306) because the device library consumes them
as-is.
"""
import os
import math
import numpy

try:  # pragma: no cover - Cassiopee is optional
    import Converter.Internal as Internal
    import Converter.PyTree as _C

    def _getState(ref):
        return _C.getState(ref)
except Exception:  # stand-alone mode
    from . import Internal
    from . import pytree_lite as _C

    def _getState(ref):
        return _C.getState(ref)

# --- Parameter_int indices (Fast/FastS/FastS/param_solver.h:77-214) -------------------
NIJK = 0; NIJK_MTR = 5; NIJK_XYZ = 10; NIJK_VENT = 15; IJKV = 20
NDIMD_SDM = 23; MXSSDOM_LU = 24; ITYPZONE = 25; ITYPVENT = 26; IFLOW = 27; ILES = 28
ITYPCP = 29; SIZE_SSDOM = 30; KFLUDOM = 33; LALE = 34; IFLAGFLT = 35
NEQ = 36; NEQ_IJ = 37; NEQ_K = 38; NEQ_COE = 39; NEQ_VENT = 40
NDIMDX = 41; NDIMDX_XYZ = 42; NDIMDX_MTR = 43; NDIMDX_VENT = 44
IO_THREAD = 45; DTLOC = 46; SA_INT = 47; RK = 52; LEVEL = 53; EXPLOC = 54
EXPLOCTYPE = 55; LEVELG = 56; LEVELD = 57; NSSITER = 58
CACHEBLCKI = 59; CACHEBLCKJ = 60; CACHEBLCKK = 61; SFD = 62; SFD_INIT_IT = 63
SLOPE = 64; NIT_INFLOW = 65; SHIFTVAR = 66; EXTRACT_RES = 67; SA_DEBUG = 68
PT_OMP = 69; PT_BC = 70; NB_RELAX = 71; NB_RESTART = 72; NB_KRYLOV = 73
IMPLICITSOLVER = 74; LU_MATCH = 75; IBC = 76; SRC = 83; MESHTYPE = 84
SENSORTYPE = 85; SCHEDULER = 86; WM_FUNCTION = 87; WM_SAMPLING = 88
NEQ_LBM = 89; LBM_COL_OP = 92; LBM_SPONGE = 102
IBC_PT_FLUX = 129; SA_LOW_RE = 130; SA_ROT_CORR = 131; SA_DIST = 132
PT_INTERP = 133; NONZ = 134; PT_VOL = 135
BC_TYPE = 0; BC_IDIR = 1; BC_FEN = 2; BC_NBDATA = 8
# --- Parameter_real indices (param_solver.h:225-306) ----------------------------------
DTC = 0; GAMMA = 1; CVINF = 2; ROINF = 3; PINF = 4; VINF = 5; TINF = 6; MINF = 7
RONUTILDEINF = 8; REYNOLDS = 9; PRANDT = 10; XMUL0 = 11; TEMP0 = 12; CS = 13
EPSI_NEWTON = 14; CFL = 15; SA_REAL = 16; VXINF = 19; VYINF = 20; VZINF = 21
TEMPS = 22; PSIROE = 34; EPSI_INFLOW = 37; PRANDT_TUR = 38; EPSI_LINEAR = 39
HPC_CUPS = 40

# Boundary-condition tags (Fast/FastC/FastC/PyTree.py:30-82)
tagBCDict = {
    "BCExtrapolate": 0, "BCDegenerateLine": 0, "BCFarfield": 1, "BCInflowSupersonic": 2,
    "BCWallInviscid": 3, "BCSymmetryPlane": 3, "BCWall": 4, "FamilySpecified": 5,
    "BCWallViscous": 6, "BCWallViscous_isot_fich": 7, "BC_fich": 8, "Nearmatch": 9,
    "BCOutflow": 10, "BCautoperiod": 11, "BCWallViscous_transition": 12, "BCInflow": 13,
    "BCExtrapolateRANS": 14, "BCPeriodic": 15, "BCOutpres": 16, "BCInj1": 17,
    "BCRacinf": 18, "BCInflowLund": 19, "BCInjMFR": 20, "BCOutMFR": 21, "BCOverlap": 22,
    "BCWallModel": 30, "BCWallExchange": 31, "BCWallViscousIsothermal": 32,
}

varsN = ['Density']
varsP = ['Density_P1']
varsM = ['Density_M1']

# The device engine has no host thread team.  The one place where the reference's result depends on its number of
# OpenMP threads is the LU-SGS solve of the implicit scheme, which every thread runs on its own block of a zone
# (lu_thread_blocks below): OMP_NUM_THREADS > 1 asks the device for the result of the reference run with that many
# threads (and gives it that many independent blocks to sweep side by side); 1 = one sweep per zone.
OMP_NUM_THREADS = int(os.environ.get('FASTS_HIP_LU_THREADS', '1'))
MX_SYNCHRO = 1000
MX_SSZONE = 10
MX_OMP_SIZE_INT = -1
OMP_MODE = 0
FIRST_IT = 0
HOOK = None


def tagBC(bcname):
    tag = tagBCDict.get(bcname, -1)
    if tag == -1:
        print("Warning: Fast: unknown BC type %s." % bcname)
    return tag


# ======================================================================================
def _setNum2Zones(a, num):
    """Set numeric data dictionary in zones (Fast/FastC/FastC/PyTree.py:113-121)."""
    zones = Internal.getNodesFromType2(a, 'Zone_t')
    for z in zones:
        cont = Internal.createUniqueChild(z, '.Solver#define', 'UserDefinedData_t')
        for k in num:
            Internal.createUniqueChild(cont, k, 'DataArray_t', num[k])
    return None


def setNum2Zones(a, num):
    ap = Internal.copyRef(a)
    _setNum2Zones(ap, num)
    return ap


def _setNum2Base(a, num):
    """Set numeric data dictionary in bases (Fast/FastC/FastC/PyTree.py:130-135).
    As in the reference, the node is attached to ``a`` itself (the tree root)."""
    cont = Internal.createUniqueChild(a, '.Solver#define', 'UserDefinedData_t')
    for k in num:
        Internal.createUniqueChild(cont, k, 'DataArray_t', num[k])
    return None


def setNum2Base(a, num):
    ap = Internal.copyRef(a)
    _setNum2Base(ap, num)
    return ap


def _reorder(t, tc=None):
    """Order the zones of tc like those of t (Fast/FastC/FastC/PyTree.py:140-210; the
    OpenMP cost sort of the reference is irrelevant on the device and is not applied)."""
    if tc is None:
        return None
    for b, bc in zip(Internal.getBases(t), Internal.getBases(tc)):
        names = [z[0] for z in Internal.getZones(b)]
        zc = {z[0]: z for z in Internal.getZones(bc)}
        others = [c for c in bc[2] if c[3] != 'Zone_t']
        bc[2] = others + [zc[n] for n in names if n in zc]
    return None


def _get(d, key, default):
    if d is None:
        return default
    a = Internal.getNodeFromName1(d, key)
    if a is None:
        return default
    return Internal.getValue(a)


def lssiter_loc(zones):
    """lssiter_loc of the reference's work dictionary (Fast/FastC/FastC/PyTree.py:2062-2076, read on the first zone):
    1 = 'implicit_local', Newton sub-iterations on the sub-zones that have not converged; 0 otherwise."""
    a = Internal.getNodeFromName2(zones[0], 'temporal_scheme') if zones else None
    return 1 if a is not None and Internal.getValue(a) == 'implicit_local' else 0


# ======================================================================================
def _buildOwnData(t, Padding=None):
    """Create .Solver#ownData (Parameter_int / Parameter_real) on every zone and
    .Solver#dtloc on the tree.  Follows Fast/FastC/FastC/PyTree.py:918-2053 for the keys
    the structured Navier-Stokes time step reads."""
    global MX_OMP_SIZE_INT
    zones = Internal.getZones(t)
    d = Internal.getNodeFromName1(t, '.Solver#define')
    temporal_scheme = _get(d, 'temporal_scheme', 'implicit')
    ss_iteration = int(_get(d, 'ss_iteration', 30))
    rk = int(_get(d, 'rk', 3))
    # 'implicit_local' changes the default period of the verification steps (Fast/FastC/FastC/PyTree.py:1134)
    modulo_verif = int(_get(d, 'modulo_verif', 7 if temporal_scheme == 'implicit_local' else 200))
    restart_fields = int(_get(d, 'restart_fields', 1))
    ompmode = int(_get(d, 'omp_mode', OMP_MODE))
    exploc = 0
    maxlevel = 1
    if temporal_scheme in ('implicit', 'implicit_local'):
        rk = 3
    if temporal_scheme == 'explicit':
        nssiter = {1: 1, 2: 2, 3: 3}.get(rk, 3)
    elif temporal_scheme in ('implicit', 'implicit_local'):
        nssiter = ss_iteration + 1
    else:
        raise ValueError("fast_amd: temporal_scheme '%s' is not supported "
                         "(explicit | implicit)." % temporal_scheme)

    timelevel_motion = 0; timelevel_target = 0; timelevel_prfile = 0; it0 = 0
    first = Internal.getNodeFromName1(t, 'Iteration')
    if first is not None:
        it0 = int(Internal.getValue(first))
    temps = 0.
    first = Internal.getNodeFromName1(t, 'Time')
    if first is not None:
        temps = float(Internal.getValue(first))

    o = Internal.createUniqueChild(t, '.Solver#ownData', 'UserDefinedData_t')
    ntask = len(zones) * 2
    size_task = (6 + 7 * OMP_NUM_THREADS) * ntask
    if MX_OMP_SIZE_INT < 1 + nssiter * (2 + size_task):   # grows with the tree at hand
        MX_OMP_SIZE_INT = 1 + nssiter * (2 + size_task)
    size_omp = MX_OMP_SIZE_INT
    dtdim = 12 + 1
    datap = numpy.zeros(dtdim + size_omp, Internal.E_NpyInt)
    datap[0] = nssiter
    datap[1] = modulo_verif
    datap[2] = restart_fields - 1
    datap[3] = timelevel_motion
    datap[4] = timelevel_target
    datap[5] = timelevel_prfile
    datap[6] = rk
    datap[7] = it0
    datap[8] = ompmode
    datap[9] = maxlevel
    datap[10] = 0
    datap[11] = dtdim
    datap[12] = 1
    datap[dtdim:dtdim + nssiter] = -1
    Internal.createUniqueChild(o, '.Solver#dtloc', 'DataArray_t', datap)

    ndom = 0
    for b in Internal.getBases(t):
        for z in Internal.getZones(b):
            oz = Internal.createUniqueChild(z, '.Solver#ownData', 'UserDefinedData_t')
            dz = Internal.getNodeFromName1(z, '.Solver#define')
            model = 'Euler'
            a = Internal.getNodeFromName2(z, 'GoverningEquations')
            if a is None:
                a = Internal.getNodeFromName2(b, 'GoverningEquations')
            if a is not None:
                model = Internal.getValue(a)
            ref = Internal.getNodeFromName1(z, 'ReferenceState')
            if ref is None:
                ref = Internal.getNodeFromName1(b, 'ReferenceState')
            adim = None
            prandtltb = 1.
            if ref is not None:
                adim = _getState(ref)
                prandtltb = adim[18]
            else:
                print('Warning: Fast: can not find a refState.')

            scheme = _get(dz, 'scheme', 'ausmpred')
            slope = _get(dz, 'slope', 'o3')
            motion = _get(dz, 'motion', 'none')
            dtnature = _get(dz, 'time_step_nature', 'global')
            dtc = float(_get(dz, 'time_step', -0.000001))
            epsi_newton = float(_get(dz, 'epsi_newton', 0.1))
            epsi_linear = float(_get(dz, 'epsi_linear', 0.01))
            cfl = float(_get(dz, 'cfl', 1.))
            psiroe = float(_get(dz, 'psiroe', 0.1))
            prandtltb = float(_get(dz, 'prandtltb', prandtltb))
            ratiom = float(_get(dz, 'ratiom', 10000.))
            senseurtype = int(_get(dz, 'senseurType', 1))
            cacheblckI = int(_get(dz, 'cache_blocking_I', 2048))
            cacheblckJ = int(_get(dz, 'cache_blocking_J', 2))
            cacheblckK = int(_get(dz, 'cache_blocking_K', 7))
            shiftvar = int(_get(dz, 'shiftvar', 0))
            io_th = int(_get(dz, 'io_thread', 0))
            nbr_relax = int(_get(dz, 'nb_relax', 1))
            lu_match = int(_get(dz, 'lu_match', 0))
            extract_res = int(_get(dz, 'extract_res', 0))
            sgsmodel = _get(dz, 'sgsmodel', 'Miles')
            implicit_solver = _get(dz, 'implicit_solver', 'lussor')
            sfd = int(_get(dz, 'sfd', 0))
            sfd_chi = float(_get(dz, 'sfd_chi', 0.))
            sfd_delta = float(_get(dz, 'sfd_delta', 1.e15))
            sfd_init_iter = int(_get(dz, 'sfd_init_iter', 1))
            nit_inflow = int(_get(dz, 'inj1_newton_nit', 10))
            epsi_inflow = float(_get(dz, 'inj1_newton_tol', 1.e-5))

            sa_dist = 0
            sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
            if sol is not None and Internal.getNodeFromName1(sol, 'TurbulentDistance') is not None:
                sa_dist = 1

            iflow = 1
            ides = 0; idist = 1; ispax = 2; izgris = 0; iprod = 0
            azgris = 0.01; addes = 0.2
            if model in ('Euler', 'euler'): iflow = 1
            elif model in ('NSLaminar', 'nslaminar', 'NSLes', 'nsles'): iflow = 2
            elif model in ('nsspalart', 'NSTurbulent'): iflow = 3
            else:
                raise ValueError('fast_amd: model %s is invalid.' % model)
            iles = 0
            if sgsmodel != 'Miles':
                raise ValueError("fast_amd: sgsmodel '%s' is not supported (Miles only)." % sgsmodel)
            if motion != 'none':
                raise ValueError("fast_amd: motion '%s' is not supported (none only)." % motion)
            lale = 0
            itypcp = 2 if temporal_scheme == 'explicit' else 1
            # blocks of the Newton residual; with 'implicit_local' also the grain of the sub-zones
            # (Fast/FastC/FastC/PyTree.py:1305,1678-1694)
            size_ssdom = [100, 10, 10]
            if temporal_scheme == 'implicit_local':
                size_ssdom = [int(v) for v in numpy.asarray(_get(dz, 'ssdom_IJK', [240, 20, 900])).reshape(-1)[:3]]
                if len(size_ssdom) != 3 or min(size_ssdom) < 1:
                    raise ValueError("fast_amd: ssdom_IJK must hold three positive block sizes.")
            islope = {'o1': 1, 'o2': 4, 'o3': 2, 'o5': 5, 'minmod': 3, 'o3sc': 6, 'o5sc': 7}.get(slope, 2)
            if scheme == 'ausmpred': kfludom = 1
            elif scheme == 'senseur': kfludom = 2
            elif scheme == 'roe_kap': kfludom = 5; islope = 2
            elif scheme == 'roe_min': kfludom = 5; islope = 3
            elif scheme == 'roe_nul': kfludom = 5; islope = 1
            elif scheme == 'roe': kfludom = 5
            else:
                raise ValueError('fast_amd: scheme %s is invalid.' % scheme)
            dtloc = {'global': 0, 'local': 1}[dtnature]
            ImplicitSolverNum = {'lussor': 0, 'gmres': 1}[implicit_solver]

            pint = numpy.empty(137, Internal.E_NpyInt)
            pint[:] = -1000
            pint[0:25] = -1
            pint[25] = 0; pint[26] = 0
            pint[27] = iflow; pint[28] = iles; pint[29] = itypcp
            pint[30:33] = size_ssdom
            pint[33] = kfludom; pint[34] = lale; pint[35] = 0
            pint[36:45] = -1
            pint[45] = io_th; pint[46] = dtloc
            pint[47] = ides; pint[48] = idist; pint[49] = ispax; pint[50] = izgris; pint[51] = iprod
            pint[52] = rk; pint[53] = 1; pint[54] = exploc; pint[55] = 0; pint[56] = 1; pint[57] = 0
            pint[58] = nssiter
            pint[59] = cacheblckI; pint[60] = cacheblckJ; pint[61] = cacheblckK
            pint[62] = sfd; pint[63] = sfd_init_iter; pint[64] = islope; pint[65] = nit_inflow
            pint[66] = shiftvar; pint[67] = extract_res; pint[68] = 0
            pint[69] = 0; pint[70] = 0
            pint[71] = nbr_relax; pint[72] = 1; pint[73] = 20
            pint[74] = ImplicitSolverNum; pint[75] = lu_match
            pint[76:83] = 0
            pint[83] = 0; pint[84] = 1; pint[85] = senseurtype; pint[86] = -1
            pint[87] = 0; pint[88] = 4
            pint[89:129] = 0
            pint[NEQ_LBM] = 19; pint[LBM_COL_OP] = 1
            pint[105:109] = 1
            pint[IBC_PT_FLUX] = -1
            pint[SA_LOW_RE] = 0; pint[SA_ROT_CORR] = 0; pint[SA_DIST] = sa_dist
            pint[NONZ] = ndom
            Internal.createUniqueChild(oz, 'Parameter_int', 'DataArray_t', pint)

            preal = numpy.zeros(72, numpy.float64)
            if dtc < 0:
                print('Warning: Fast: time_step set to default value (0.0001).')
                dtc = 0.0001
            preal[0] = dtc
            if adim is not None:
                preal[1] = adim[11]; preal[2] = adim[7]; preal[3] = adim[0]; preal[4] = adim[5]
                preal[5] = math.sqrt(adim[1] ** 2 + adim[2] ** 2 + adim[3] ** 2) / adim[0]
                preal[6] = adim[6]; preal[7] = adim[8]; preal[8] = adim[14]; preal[9] = adim[9]
                preal[10] = adim[18]; preal[11] = adim[15]; preal[12] = adim[17]; preal[13] = adim[16]
                preal[19] = adim[1] / adim[0]; preal[20] = adim[2] / adim[0]; preal[21] = adim[3] / adim[0]
            preal[14] = epsi_newton; preal[15] = cfl
            preal[16] = azgris; preal[17] = addes; preal[18] = ratiom
            preal[22] = temps
            preal[34] = psiroe; preal[35] = sfd_chi; preal[36] = sfd_delta
            preal[37] = epsi_inflow; preal[38] = prandtltb; preal[39] = epsi_linear
            preal[40] = 1.
            preal[42] = 1. / math.sqrt(3.); preal[43] = 0.5; preal[51] = 0.985; preal[52] = 1.
            preal[55] = 0.009; preal[56] = 0.015
            preal[60] = numpy.sqrt(((0.25 * 0.1) ** 2 + 1)) - 0.25 * 0.1
            preal[61] = 0.1; preal[62] = 0.1; preal[63] = 0.1
            preal[65] = 1.
            preal[66:72] = [0., 1., 1., 1., 0., 0.]
            Internal.createUniqueChild(oz, 'Parameter_real', 'DataArray_t', preal)

            Internal.createUniqueChild(oz, 'CFL_minmaxmoy', 'DataArray_t',
                                       numpy.zeros(3, numpy.float64))
            Internal.createUniqueChild(oz, 'type_zone', 'DataArray_t', 0)
            Internal.createUniqueChild(oz, 'model', 'DataArray_t', model)
            Internal.createUniqueChild(oz, 'temporal_scheme', 'DataArray_t', temporal_scheme)
            Internal.createUniqueChild(oz, 'exp_local', 'DataArray_t', exploc)
            Internal.createUniqueChild(oz, 'rk', 'DataArray_t', rk)
            ndom += 1
    return None


def _build_omp(t):
    """Reserve the (empty, for structured zones) thread-map area at the end of
    Parameter_int and record its offset (Fast/FastC/FastC/PyTree.py:870-913)."""
    for z in Internal.getZones(t):
        o = Internal.getNodeFromName1(z, '.Solver#ownData')
        p = Internal.getNodeFromName1(o, 'Parameter_int')
        c = p[1].size
        datap = numpy.zeros(c, Internal.E_NpyInt)
        datap[:c] = p[1][:c]
        datap[PT_OMP] = c
        datap[SCHEDULER] = c
        p[1] = datap
    return None


def lu_thread_blocks(pint, nthreads, lmin=10):
    """Blocks of interior cells that the reference's OpenMP threads take in a zone (omp_mode 0: every zone is cut among
    all the threads, Fast/FastC/FastC/HPC_LAYER/distributeThreads.cpp:90-95 -> indice_boucle_lu.for:94-335), as an
    (n, 6) int32 array of (imin, imax, jmin, jmax, kmin, kmax), cells numbered from 1; threads left without cells are
    dropped.  The cut is p x q with p a divisor of nthreads not larger than 8, along k and j when both directions
    offer enough slices of lmin cells (10 with the implicit scheme), else along j and i; the pair with the smallest
    load imbalance wins (an uneven cut along the second direction counts 2 or 5 times)."""
    n = int(nthreads)
    size = [int(pint[IJKV]), int(pint[IJKV + 1]), int(pint[IJKV + 2])]
    whole = numpy.array([[1, size[0], 1, size[1], 1, size[2]]], dtype=numpy.int32)
    if n <= 1:
        return whole
    slices = [max(1, v // lmin) for v in size]
    if size[2] == 1:
        slices[2] = 1
    divisors = [1] + [q for q in range(2, 9) if n % q == 0]
    f32 = numpy.float32

    def imbalance(a, ca, b, cb, weight):     # direction a cut in ca, direction b in cb
        return float(f32(size[a] % ca) / f32(size[a] // ca)) + float(f32(size[b] % cb) / f32(size[b] // cb)) * weight

    def best_cut(a, b, weight, small_first):
        """(count along a, count along b) or None; a = the direction the reference prefers to cut.  small_first: the
        pair with the divisor along a is tried before the pair with it along b (a later pair wins only when cheaper)."""
        best, cost_best = None, 100000.
        for p1 in divisors:
            p2 = n // p1
            for (ca, cb) in (((p1, p2), (p2, p1)) if small_first else ((p2, p1), (p1, p2))):
                if ca <= slices[a] and cb <= slices[b]:
                    cost = imbalance(a, ca, b, cb, weight)
                    if cost < cost_best:
                        best, cost_best = (ca, cb), cost
        return best

    counts = None
    failed = False
    if slices[2] * slices[1] >= n:
        cut = best_cut(2, 1, 2., False)
        if cut is not None:
            counts = [1, cut[1], cut[0]]
        else:
            failed = True
    if counts is None and slices[0] * slices[1] >= n:
        cut = best_cut(1, 0, 5., True)
        if cut is not None:
            counts = [cut[1], cut[0], 1]
        else:
            failed = True
    if counts is None:
        if failed:
            return whole                      # the first thread takes the zone
        # not enough slices for every thread: as many blocks as there are slices, the other threads idle
        if slices[0] * slices[1] > slices[1] * slices[2]:
            counts = [slices[0], slices[1], 1]
        else:
            counts = [1, slices[1], slices[2]]

    def part(d, q):                            # q-th (from 0) of counts[d] parts of direction d
        c, v = counts[d], size[d]
        sz, rest = v // c, v % c
        if q < rest:
            return 1 + (sz + 1) * q, (sz + 1) * (q + 1)
        return 1 + sz * q + rest, sz * (q + 1) + rest

    blocks = []
    if counts[0] == 1:
        for t in range(min(n, counts[1] * counts[2])):
            kq, jq = divmod(t, counts[1])
            blocks.append((1, size[0]) + part(1, jq) + part(2, kq))
    else:
        for t in range(min(n, counts[0] * counts[1])):
            jq, iq = divmod(t, counts[0])
            blocks.append(part(0, iq) + part(1, jq) + (1, size[2]))
    return numpy.array(blocks, dtype=numpy.int32)


# ======================================================================================
def degenerate_cells(x, y, z):
    """Cells collapsed by the ghost-cell construction at a boundary: the four edges along one direction have zero
    length (Fast/FastC/FastC/Metric/nature_geom_dom.for:52-180, tolerance 2e-8).  x, y, z are node arrays
    (ni, nj, nk); returns a bool array (ni-1, nj-1, nk-1) indexed like the cells."""
    eps = 2.e-8
    ni, nj, nk = x.shape

    def seg(a, b):
        return numpy.sqrt((x[a] - x[b]) ** 2 + (y[a] - y[b]) ** 2 + (z[a] - z[b]) ** 2)

    def sl(di, dj, dk):
        return (slice(di, ni - 1 + di), slice(dj, nj - 1 + dj), slice(dk, nk - 1 + dk))
    deg = numpy.zeros((ni - 1, nj - 1, nk - 1), dtype=bool)
    dirs = [
        [((0, 0, 0), (1, 0, 0)), ((0, 1, 0), (1, 1, 0)), ((0, 1, 1), (1, 1, 1)), ((0, 0, 1), (1, 0, 1))],
        [((0, 0, 0), (0, 1, 0)), ((1, 0, 0), (1, 1, 0)), ((1, 0, 1), (1, 1, 1)), ((0, 0, 1), (0, 1, 1))],
        [((0, 0, 0), (0, 0, 1)), ((1, 0, 0), (1, 0, 1)), ((1, 1, 0), (1, 1, 1)), ((0, 1, 0), (0, 1, 1))],
    ]
    for edges in dirs:
        allz = numpy.ones_like(deg)
        for (a, b) in edges:
            allz &= seg(sl(*a), sl(*b)) <= eps
        deg |= allz
    return deg


def _ghost_planes(pint, typ):
    """(first interior, last interior, ghost depth) per direction and the face order of tijk_extrap.for /
    dist_extrap.for: kmin, kmax (when the zone has ghost layers along k), jmin, jmax, imin, imax."""
    hi = [int(pint[IJKV]), int(pint[IJKV + 1]), int(pint[IJKV + 2]) if typ != 3 else 1]
    nf = [int(pint[NIJK + 3]), int(pint[NIJK + 3]), int(pint[NIJK + 4])]
    faces = ([(2, 0), (2, 1)] if nf[2] != 0 else []) + [(1, 0), (1, 1), (0, 0), (0, 1)]
    return hi, nf, faces


def tijk_extrap(pint, deg, mtr):
    """Metrics of the collapsed ghost cells (Fast/FastC/FastC/Metric/tijk_extrap.for, called by init_metric.cpp:318):
    the surface vectors of the transverse faces and the volume are taken from the mirror cell inside the zone,
    and the vector of the face along the normal is rebuilt so that the six vectors of the cell still sum to zero.
    Two passes, the second one over the edge and corner ghost cells too.  In place on the flat metric buffer
    [ti | tj | tk | vol] of a curvilinear (type 0) or two-dimensional (type 3) zone; deg from degenerate_cells."""
    typ = int(pint[ITYPZONE])
    if typ not in (0, 3) or not deg.any():
        return
    ni, nj, nk = (int(v) for v in pint[NIJK_XYZ:NIJK_XYZ + 3])
    nkm = nk if typ == 0 else 1
    n = int(pint[NDIMDX_MTR]); neq_ij = int(pint[NEQ_IJ]); neq_k = int(pint[NEQ_K])
    neq = [neq_ij, neq_ij, neq_k]
    off = [0, n * neq_ij, 2 * n * neq_ij]
    T = [[mtr[off[d] + c * n:off[d] + (c + 1) * n].reshape((ni, nj, nkm), order='F') for c in range(neq[d])]
         for d in range(3)]
    vol = mtr[2 * n * neq_ij + n * neq_k:].reshape((ni, nj, nkm), order='F')
    hi, nf, faces = _ghost_planes(pint, typ)

    def at(idx):      # cell / face indices (1 = first interior) -> array indices
        return tuple(idx[d] + nf[d] - 1 for d in range(3))

    for npass in (1, 2):
        lo_l = [1 if npass == 1 else 1 - nf[d] for d in range(3)]
        hi_l = [hi[d] if npass == 1 else hi[d] + nf[d] for d in range(3)]
        for (d, side) in faces:
            a, b = [t for t in range(3) if t != d]
            planes = range(0, -nf[d], -1) if side == 0 else range(hi[d] + 1, hi[d] + nf[d] + 1)
            A, B = numpy.meshgrid(numpy.arange(lo_l[a], hi_l[a] + 1), numpy.arange(lo_l[b], hi_l[b] + 1), indexing='ij')
            for g in planes:
                g0 = 1 - g if side == 0 else 2 * hi[d] + 1 - g
                cell = [None] * 3; cell[d] = numpy.full_like(A, g); cell[a] = A; cell[b] = B
                mir = list(cell); mir[d] = numpy.full_like(A, g0)
                M = deg[at(cell)]
                if not M.any():
                    continue

                def shifted(idx, t, by=1):
                    out = list(idx); out[t] = idx[t] + by
                    return out
                for t in (a, b):
                    if neq[t] == 0:
                        continue
                    nb = list(cell); nb[t] = numpy.maximum(1 - nf[t], cell[t] - 1)
                    c1 = M & deg[at(nb)]
                    nb[t] = numpy.minimum(hi[t] + nf[t], cell[t] + 1)
                    c2 = M & deg[at(nb)]
                    for c in range(neq[t]):
                        w = tuple(v[c1] for v in at(cell)); r = tuple(v[c1] for v in at(mir))
                        T[t][c][w] = T[t][c][r]
                        w = tuple(v[c2] for v in at(shifted(cell, t))); r = tuple(v[c2] for v in at(shifted(mir, t)))
                        T[t][c][w] = T[t][c][r]
                # vector of the face along the normal (same operand order as the reference)
                sel = lambda idx: tuple(v[M] for v in at(idx))
                c0, cd = sel(cell), sel(shifted(cell, d))
                if neq_k == 3:
                    ca, cb = sel(shifted(cell, a)), sel(shifted(cell, b))
                    for c in range(neq_ij):
                        if side == 0:
                            T[d][c][c0] = T[d][c][cd] - T[a][c][c0] + T[a][c][ca] - T[b][c][c0] + T[b][c][cb]
                        else:
                            T[d][c][cd] = T[d][c][c0] + T[a][c][c0] - T[a][c][ca] + T[b][c][c0] - T[b][c][cb]
                elif d == 2:
                    for c in range(neq_k):
                        if side == 0: T[2][c][c0] = T[2][c][cd]
                        else: T[2][c][cd] = T[2][c][c0]
                else:
                    o = 1 - d
                    co = sel(shifted(cell, o))
                    for c in range(neq_ij):
                        if side == 0:
                            T[d][c][c0] = T[d][c][cd] - T[o][c][c0] + T[o][c][co]
                        else:
                            T[d][c][cd] = T[d][c][c0] + T[o][c][c0] - T[o][c][co]
                vol[c0] = vol[sel(mir)]


def dist_extrap(pint, deg, dist):
    """Wall distance of the collapsed ghost cells = that of the mirror cell inside the zone
    (Fast/FastC/FastC/Metric/dist_extrap.for, init_metric.cpp:324-329).  dist: cell array (ni, nj, nk), in place."""
    if not deg.any():
        return
    typ = int(pint[ITYPZONE])
    hi, nf, faces = _ghost_planes(pint, typ)
    hi[2] = int(pint[IJKV + 2])

    def at(idx):
        return tuple(idx[d] + nf[d] - 1 for d in range(3))
    for npass in (1, 2):
        lo_l = [1 if npass == 1 else 1 - nf[d] for d in range(3)]
        hi_l = [hi[d] if npass == 1 else hi[d] + nf[d] for d in range(3)]
        for (d, side) in faces:
            a, b = [t for t in range(3) if t != d]
            planes = range(0, -nf[d], -1) if side == 0 else range(hi[d] + 1, hi[d] + nf[d] + 1)
            A, B = numpy.meshgrid(numpy.arange(lo_l[a], hi_l[a] + 1), numpy.arange(lo_l[b], hi_l[b] + 1), indexing='ij')
            for g in planes:
                g0 = 1 - g if side == 0 else 2 * hi[d] + 1 - g
                cell = [None] * 3; cell[d] = numpy.full_like(A, g); cell[a] = A; cell[b] = B
                mir = list(cell); mir[d] = numpy.full_like(A, g0)
                M = deg[at(cell)]
                if M.any():
                    dist[tuple(v[M] for v in at(cell))] = dist[tuple(v[M] for v in at(mir))]


# ======================================================================================
def nature_geom_dom(x, y, z, ific, ifick):
    """Zone type detection (0 curvilinear, 1 k-homogeneous, 2 Cartesian constant step,
    3 two-dimensional), vectorised restatement of
    Fast/FastC/FastC/Metric/nature_geom_dom.for:186-503 (tolerance 2e-8, :29).
    x, y, z are node arrays (ni, nj, nk).  Collapsed cells (:52-180) are ignored in the
    tests, as in the reference.  Returns (typ_zone, nijk_mtr[5], ndimdx_mtr, neq_ij, neq_k)."""
    eps = 2.e-8
    ni, nj, nk = x.shape
    if nk == 2:
        return 3, [1, ni, 0, ific, ifick], ni * nj, 2, 0
    deg = degenerate_cells(x, y, z)
    ok = ~deg
    # index (1,1,1) of the reference = first interior node = [ific, ific, ifick]
    i1, j1, k1 = ific, ific, ifick

    def plane_const(a, axis, ref_idx):
        # |a(ref line) - a(i,j,k)| and |... a(i+1,j+1 / k+1 ...)| both on plane and plane+1
        if axis == 2:
            r0 = a[i1, j1, :]
            t1 = numpy.abs(r0[None, None, :-1] - a[:-1, :-1, :-1]); t2 = numpy.abs(r0[None, None, :-1] - a[1:, 1:, :-1])
            t3 = numpy.abs(r0[None, None, 1:] - a[:-1, :-1, 1:]); t4 = numpy.abs(r0[None, None, 1:] - a[1:, 1:, 1:])
        elif axis == 1:
            r0 = a[i1, :, k1]
            t1 = numpy.abs(r0[None, :-1, None] - a[:-1, :-1, :-1]); t2 = numpy.abs(r0[None, :-1, None] - a[1:, :-1, 1:])
            t3 = numpy.abs(r0[None, 1:, None] - a[:-1, 1:, :-1]); t4 = numpy.abs(r0[None, 1:, None] - a[1:, 1:, 1:])
        else:
            r0 = a[:, j1, k1]
            t1 = numpy.abs(r0[:-1, None, None] - a[:-1, :-1, :-1]); t2 = numpy.abs(r0[:-1, None, None] - a[:-1, 1:, 1:])
            t3 = numpy.abs(r0[1:, None, None] - a[1:, :-1, :-1]); t4 = numpy.abs(r0[1:, None, None] - a[1:, 1:, 1:])
        bad = ((t1 >= eps) | (t2 >= eps) | (t3 >= eps) | (t4 >= eps)) & ok
        return not bad.any()

    def step_const(a, axis):
        if axis == 2:
            cible = a[i1, j1, k1 + 1] - a[i1, j1, k1]; dd = a[:-1, :-1, 1:] - a[:-1, :-1, :-1]
        elif axis == 1:
            cible = a[i1, j1 + 1, k1] - a[i1, j1, k1]; dd = a[:-1, 1:, :-1] - a[:-1, :-1, :-1]
        else:
            cible = a[i1 + 1, j1, k1] - a[i1, j1, k1]; dd = a[1:, :-1, :-1] - a[:-1, :-1, :-1]
        bad = (numpy.abs(dd - cible) >= eps) & ok
        return not bad.any()

    zcart = plane_const(z, 2, 0); zcte = step_const(z, 2)
    ycart = plane_const(y, 1, 0); ycte = step_const(y, 1)
    xcart = plane_const(x, 0, 0); xcte = step_const(x, 0)
    if xcart and xcte and ycart and ycte and zcart and zcte:
        return 2, [0, 0, 0, 0, 0], 1, 1, 1
    if zcart and zcte:
        def kinv(a):
            r = a[:, :, k1]
            t1 = numpy.abs(r[:-1, :-1, None] - a[:-1, :-1, :-1]); t2 = numpy.abs(r[1:, 1:, None] - a[1:, 1:, :-1])
            t3 = numpy.abs(r[:-1, :-1, None] - a[:-1, :-1, 1:]); t4 = numpy.abs(r[1:, 1:, None] - a[1:, 1:, 1:])
            bad = ((t1 >= eps) | (t2 >= eps) | (t3 >= eps) | (t4 >= eps)) & ok
            return not bad.any()
        if kinv(x) and kinv(y):
            return 1, [1, ni, 0, ific, 0], ni * nj, 2, 1
    return 0, [1, ni, ni * nj, ific, ifick], ni * nj * nk, 3, 3


def cart_metrics(x, y, z, ific, ifick):
    """ti, tj, tk, vol of a Cartesian zone, evaluated on the first interior cell exactly as
    Fast/FastC/FastC/Metric/cp_tijk.for:295-321 and cp_vol.for:150-172 do (same operand
    order, so the result is bit-identical to the reference's)."""
    def X(a, di, dj, dk):
        return float(a[ific + di, ific + dj, ifick + dk])
    def p(a, n):  # node n of the cell, numbered l0..l7 as in the reference
        d = {0: (0, 0, 0), 1: (0, 1, 0), 2: (0, 0, 1), 3: (0, 1, 1),
             4: (1, 0, 0), 5: (1, 1, 0), 6: (1, 0, 1), 7: (1, 1, 1)}[n]
        return X(a, *d)
    tix10 = (p(y, 1) - p(y, 0)) * (p(z, 3) - p(z, 0)) - (p(z, 1) - p(z, 0)) * (p(y, 3) - p(y, 0))
    tix11 = (p(y, 2) - p(y, 3)) * (p(z, 0) - p(z, 2)) - (p(z, 2) - p(z, 3)) * (p(y, 0) - p(y, 2))
    ti = .5 * (tix10 + tix11)
    tjy10 = (p(z, 0) - p(z, 4)) * (p(x, 2) - p(x, 0)) - (p(x, 0) - p(x, 4)) * (p(z, 2) - p(z, 0))
    tjy11 = (p(z, 6) - p(z, 2)) * (p(x, 4) - p(x, 6)) - (p(x, 6) - p(x, 2)) * (p(z, 4) - p(z, 6))
    tj = .5 * (tjy10 + tjy11)
    tkz10 = (p(x, 4) - p(x, 0)) * (p(y, 5) - p(y, 4)) - (p(y, 4) - p(y, 0)) * (p(x, 5) - p(x, 4))
    tkz11 = (p(x, 1) - p(x, 5)) * (p(y, 0) - p(y, 1)) - (p(y, 1) - p(y, 5)) * (p(x, 0) - p(x, 1))
    tk = .5 * (tkz10 + tkz11)
    ax = .25 * ((p(x, 4) + p(x, 5) + p(x, 7) + p(x, 6)) * ti - (p(x, 0) + p(x, 1) + p(x, 3) + p(x, 2)) * ti)
    ay = .25 * ((p(y, 1) + p(y, 3) + p(y, 5) + p(y, 7)) * tj - (p(y, 0) + p(y, 2) + p(y, 4) + p(y, 6)) * tj)
    az = .25 * ((p(z, 2) + p(z, 3) + p(z, 6) + p(z, 7)) * tk - (p(z, 0) + p(z, 1) + p(z, 4) + p(z, 5)) * tk)
    vol = max((ax + ay + az) / 3., 1.e-30)
    return ti, tj, tk, vol


def general_metrics(x, y, z):
    """ti, tj, tk (surface vectors of the i-, j-, k-min face of every cell, two-triangle average)
    and vol of a curvilinear zone, ghost cells included, with the operand order of
    Fast/FastC/FastC/Metric/cp_tijk.for:60-123 and cp_vol.for:55-93.  Arrays are dimensioned on the
    nodes (ni,nj,nk) like the reference's (index = i + j*ni + k*ni*nj, NDIMDX_MTR = ni*nj*nk);
    entries without a face or cell behind them stay 0.  Returns the flat buffer
    [ti(:,1..3) | tj(:,1..3) | tk(:,1..3) | vol]."""
    ni, nj, nk = x.shape
    n = ni * nj * nk
    X = [numpy.asarray(a, dtype=numpy.float64) for a in (x, y, z)]

    def N(a, di, dj, dk, si, sj, sk):
        """node a(i+di, j+dj, k+dk) for i in si etc."""
        return a[si.start + di:si.stop + di, sj.start + dj:sj.stop + dj, sk.start + dk:sk.stop + dk]

    def cross(p1, p0, p3, q0, c):
        # component c of (p1-p0) x (p3-q0) written as in the reference:
        #   tix = (y1-y0)*(z3-z0) - (z1-z0)*(y3-y0)
        a, b = (c + 1) % 3, (c + 2) % 3
        return (p1[a] - p0[a]) * (p3[b] - q0[b]) - (p1[b] - p0[b]) * (p3[a] - q0[a])

    T = [numpy.zeros((3, ni, nj, nk)) for _ in range(3)]
    full = slice(0, ni), slice(0, nj), slice(0, nk)
    # I faces: nodes l0=(i,j,k) l1=(i,j+1,k) l2=(i,j,k+1) l3=(i,j+1,k+1); all i, cells in j,k
    si, sj, sk = slice(0, ni), slice(0, nj - 1), slice(0, nk - 1)
    l0 = [N(a, 0, 0, 0, si, sj, sk) for a in X]; l1 = [N(a, 0, 1, 0, si, sj, sk) for a in X]
    l2 = [N(a, 0, 0, 1, si, sj, sk) for a in X]; l3 = [N(a, 0, 1, 1, si, sj, sk) for a in X]
    for c in range(3):
        t10 = cross(l1, l0, l3, l0, c)          # (l1-l0) x (l3-l0)
        t11 = cross(l2, l3, l0, l2, c)          # (l2-l3) x (l0-l2)
        T[0][c][si, sj, sk] = .5 * (t10 + t11)
    # J faces: l0, l2=(i,j,k+1), l4=(i+1,j,k), l6=(i+1,j,k+1)
    si, sj, sk = slice(0, ni - 1), slice(0, nj), slice(0, nk - 1)
    l0 = [N(a, 0, 0, 0, si, sj, sk) for a in X]; l2 = [N(a, 0, 0, 1, si, sj, sk) for a in X]
    l4 = [N(a, 1, 0, 0, si, sj, sk) for a in X]; l6 = [N(a, 1, 0, 1, si, sj, sk) for a in X]
    for c in range(3):
        t10 = cross(l0, l4, l2, l0, c)          # (l0-l4) x (l2-l0)
        t11 = cross(l6, l2, l4, l6, c)          # (l6-l2) x (l4-l6)
        T[1][c][si, sj, sk] = .5 * (t10 + t11)
    # K faces: l0, l1=(i,j+1,k), l4, l5=(i+1,j+1,k)
    si, sj, sk = slice(0, ni - 1), slice(0, nj - 1), slice(0, nk)
    l0 = [N(a, 0, 0, 0, si, sj, sk) for a in X]; l1 = [N(a, 0, 1, 0, si, sj, sk) for a in X]
    l4 = [N(a, 1, 0, 0, si, sj, sk) for a in X]; l5 = [N(a, 1, 1, 0, si, sj, sk) for a in X]
    for c in range(3):
        t10 = cross(l4, l0, l5, l4, c)          # (l4-l0) x (l5-l4)
        t11 = cross(l1, l5, l0, l1, c)          # (l1-l5) x (l0-l1)
        T[2][c][si, sj, sk] = .5 * (t10 + t11)
    # volumes (cp_vol.for:55-93)
    si, sj, sk = slice(0, ni - 1), slice(0, nj - 1), slice(0, nk - 1)
    P = {m: [N(a, (m >> 2) & 1, m & 1, (m >> 1) & 1, si, sj, sk) for a in X] for m in range(8)}
    # node numbering of the reference: l1 = +j, l2 = +k, l3 = +j+k, l4 = +i, l5 = +i+j, l6 = +i+k, l7 = all
    tiL = [T[0][c][si, sj, sk] for c in range(3)]; tiP = [T[0][c][1:ni, 0:nj - 1, 0:nk - 1] for c in range(3)]
    tjL = [T[1][c][si, sj, sk] for c in range(3)]; tjP = [T[1][c][0:ni - 1, 1:nj, 0:nk - 1] for c in range(3)]
    tkL = [T[2][c][si, sj, sk] for c in range(3)]; tkP = [T[2][c][0:ni - 1, 0:nj - 1, 1:nk] for c in range(3)]
    acc = []
    for c in range(3):
        q = {m: P[m][c] for m in range(8)}
        acc.append(.25 * ((q[4] + q[5] + q[7] + q[6]) * tiP[c]
                          - (q[0] + q[1] + q[3] + q[2]) * tiL[c]
                          + (q[1] + q[3] + q[5] + q[7]) * tjP[c]
                          - (q[0] + q[2] + q[4] + q[6]) * tjL[c]
                          + (q[2] + q[3] + q[6] + q[7]) * tkP[c]
                          - (q[0] + q[1] + q[4] + q[5]) * tkL[c]))
    vol = numpy.zeros((ni, nj, nk))
    vol[si, sj, sk] = numpy.maximum((acc[0] + acc[1] + acc[2]) / 3., 1.e-30)
    out = numpy.zeros(10 * n)
    for d in range(3):
        for c in range(3):
            out[(3 * d + c) * n:(3 * d + c + 1) * n] = T[d][c].reshape(-1, order='F')
    out[9 * n:] = vol.reshape(-1, order='F')
    return out


def planar_metrics(x, y, z):
    """ti(:,1..2), tj(:,1..2) and vol of a two-dimensional zone (zone type 3: one cell along k, no ghost layer
    along k), with the operand order of Fast/FastC/FastC/Metric/cp_tijk.for:376-402 and cp_vol.for:177-212.
    Arrays are dimensioned on the nodes of one k-plane (index = i + j*ni, NDIMDX_MTR = ni*nj).  Returns the
    flat buffer [ti(:,1..2) | tj(:,1..2) | vol]."""
    ni, nj, nk = x.shape
    assert nk == 2
    n = ni * nj
    X = [numpy.asarray(a, dtype=numpy.float64) for a in (x, y, z)]

    def N(a, di, dj, dk, si, sj):
        return a[si.start + di:si.stop + di, sj.start + dj:sj.stop + dj, dk]

    def cross(p1, p0, p3, q0, c):
        a, b = (c + 1) % 3, (c + 2) % 3
        return (p1[a] - p0[a]) * (p3[b] - q0[b]) - (p1[b] - p0[b]) * (p3[a] - q0[a])

    T = [numpy.zeros((2, ni, nj)) for _ in range(2)]
    si, sj = slice(0, ni), slice(0, nj - 1)
    l0 = [N(a, 0, 0, 0, si, sj) for a in X]; l1 = [N(a, 0, 1, 0, si, sj) for a in X]
    l2 = [N(a, 0, 0, 1, si, sj) for a in X]; l3 = [N(a, 0, 1, 1, si, sj) for a in X]
    for c in range(2):
        T[0][c][si, sj] = .5 * (cross(l1, l0, l3, l0, c) + cross(l2, l3, l0, l2, c))
    si, sj = slice(0, ni - 1), slice(0, nj)
    l0 = [N(a, 0, 0, 0, si, sj) for a in X]; l2 = [N(a, 0, 0, 1, si, sj) for a in X]
    l4 = [N(a, 1, 0, 0, si, sj) for a in X]; l6 = [N(a, 1, 0, 1, si, sj) for a in X]
    for c in range(2):
        T[1][c][si, sj] = .5 * (cross(l0, l4, l2, l0, c) + cross(l6, l2, l4, l6, c))
    si, sj = slice(0, ni - 1), slice(0, nj - 1)
    P = {m: [N(a, (m >> 2) & 1, m & 1, (m >> 1) & 1, si, sj) for a in X] for m in range(8)}
    acc = []
    for c in range(2):
        q = {m: P[m][c] for m in range(8)}
        acc.append(.25 * ((q[4] + q[5] + q[7] + q[6]) * T[0][c][1:ni, 0:nj - 1]
                          - (q[0] + q[1] + q[3] + q[2]) * T[0][c][si, sj]
                          + (q[1] + q[3] + q[5] + q[7]) * T[1][c][0:ni - 1, 1:nj]
                          - (q[0] + q[2] + q[4] + q[6]) * T[1][c][si, sj]))
    px = {m: P[m][0] for m in range(8)}; py = {m: P[m][1] for m in range(8)}; pz = {m: P[m][2] for m in range(8)}
    tkz10 = (px[4] - px[0]) * (py[5] - py[4]) - (py[4] - py[0]) * (px[5] - px[4])
    tkz11 = (px[1] - px[5]) * (py[0] - py[1]) - (py[1] - py[5]) * (px[0] - px[1])
    tk1 = .5 * (tkz10 + tkz11)
    az = .25 * (pz[2] + pz[3] + pz[6] + pz[7] - pz[0] - pz[1] - pz[4] - pz[5]) * tk1
    vol = numpy.zeros((ni, nj))
    vol[si, sj] = numpy.maximum((acc[0] + acc[1] + az) / 3., 1.e-30)
    out = numpy.zeros(5 * n)
    for d in range(2):
        for c in range(2):
            out[(2 * d + c) * n:(2 * d + c + 1) * n] = T[d][c].reshape(-1, order='F')
    out[4 * n:] = vol.reshape(-1, order='F')
    return out


def homo_metrics(x, y, z, ific, ifick):
    """ti(:,1..2), tj(:,1..2), tk(:,1) and vol of a k-homogeneous zone (zone type 1: an extruded mesh), evaluated on
    the first interior plane of cells only, with the operand order of Fast/FastC/FastC/Metric/cp_tijk.for:196-289 and
    cp_vol.for:115-146 (the k faces below and above a cell share one surface value).  Arrays are dimensioned on the
    nodes of one k-plane (index = i + j*ni, NDIMDX_MTR = ni*nj, no stride along k).  Returns the flat buffer
    [ti(:,1..2) | tj(:,1..2) | tk | vol]."""
    ni, nj, nk = x.shape
    n = ni * nj
    X = [numpy.asarray(a, dtype=numpy.float64)[:, :, ifick:ifick + 2] for a in (x, y, z)]

    def N(a, di, dj, dk, si, sj):
        return a[si.start + di:si.stop + di, sj.start + dj:sj.stop + dj, dk]

    def cross(p1, p0, p3, q0, c):
        a, b = (c + 1) % 3, (c + 2) % 3
        return (p1[a] - p0[a]) * (p3[b] - q0[b]) - (p1[b] - p0[b]) * (p3[a] - q0[a])

    T = [numpy.zeros((2, ni, nj)) for _ in range(2)]
    si, sj = slice(0, ni), slice(0, nj - 1)
    l0 = [N(a, 0, 0, 0, si, sj) for a in X]; l1 = [N(a, 0, 1, 0, si, sj) for a in X]
    l2 = [N(a, 0, 0, 1, si, sj) for a in X]; l3 = [N(a, 0, 1, 1, si, sj) for a in X]
    for c in range(2):
        T[0][c][si, sj] = .5 * (cross(l1, l0, l3, l0, c) + cross(l2, l3, l0, l2, c))
    si, sj = slice(0, ni - 1), slice(0, nj)
    l0 = [N(a, 0, 0, 0, si, sj) for a in X]; l2 = [N(a, 0, 0, 1, si, sj) for a in X]
    l4 = [N(a, 1, 0, 0, si, sj) for a in X]; l6 = [N(a, 1, 0, 1, si, sj) for a in X]
    for c in range(2):
        T[1][c][si, sj] = .5 * (cross(l0, l4, l2, l0, c) + cross(l6, l2, l4, l6, c))
    si, sj = slice(0, ni - 1), slice(0, nj - 1)
    P = {m: [N(a, (m >> 2) & 1, m & 1, (m >> 1) & 1, si, sj) for a in X] for m in range(8)}
    px = {m: P[m][0] for m in range(8)}; py = {m: P[m][1] for m in range(8)}; pz = {m: P[m][2] for m in range(8)}
    tkz10 = (px[4] - px[0]) * (py[5] - py[4]) - (py[4] - py[0]) * (px[5] - px[4])
    tkz11 = (px[1] - px[5]) * (py[0] - py[1]) - (py[1] - py[5]) * (px[0] - px[1])
    tk = numpy.zeros((ni, nj))
    tk[si, sj] = .5 * (tkz10 + tkz11)
    acc = []
    for c in range(2):
        q = {m: P[m][c] for m in range(8)}
        acc.append(.25 * ((q[4] + q[5] + q[7] + q[6]) * T[0][c][1:ni, 0:nj - 1]
                          - (q[0] + q[1] + q[3] + q[2]) * T[0][c][si, sj]
                          + (q[1] + q[3] + q[5] + q[7]) * T[1][c][0:ni - 1, 1:nj]
                          - (q[0] + q[2] + q[4] + q[6]) * T[1][c][si, sj]))
    az = .25 * ((pz[2] + pz[3] + pz[6] + pz[7]) * tk[si, sj] - (pz[0] + pz[1] + pz[4] + pz[5]) * tk[si, sj])
    vol = numpy.zeros((ni, nj))
    vol[si, sj] = numpy.maximum((acc[0] + acc[1] + az) / 3., 1.e-30)
    out = numpy.zeros(6 * n)
    for d in range(2):
        for c in range(2):
            out[(2 * d + c) * n:(2 * d + c + 1) * n] = T[d][c].reshape(-1, order='F')
    out[4 * n:5 * n] = tk.reshape(-1, order='F')
    out[5 * n:] = vol.reshape(-1, order='F')
    return out


def allocate_metric_zone(z, nssiter):
    """Fill the size/type entries of Parameter_int and return the per-zone ``metrics`` list
    ``[mtr, mtr0, vent, mtr_df, rdm, ind_dm, it_lu_ssdom, degen]``
    (Fast/FastS/FastS/Metric/allocate_metric.cpp:99-256,398-471).  For zone type 2 the
    metric array [ti,tj,tk,vol] is computed here too (fastc.init_metric's job in the
    reference, Fast/FastC/FastC/Metric/init_metric.cpp:264-322)."""
    o = Internal.getNodeFromName1(z, '.Solver#ownData')
    if o is None:
        raise ValueError("metric: numerics is missing for zone %s." % z[0])
    pint = Internal.getNodeFromName1(o, 'Parameter_int')[1]
    preal = Internal.getNodeFromName1(o, 'Parameter_real')[1]
    g = Internal.getNodeFromName1(z, 'GridCoordinates')
    if g is None:
        raise ValueError("Coord missing for metric computation")
    x = Internal.getNodeFromName1(g, 'CoordinateX')[1]
    y = Internal.getNodeFromName1(g, 'CoordinateY')[1]
    zz = Internal.getNodeFromName1(g, 'CoordinateZ')[1]
    ni, nj, nk = x.shape
    nfic = 2
    if pint[SLOPE] in (5, 7):
        raise ValueError("fast_amd: o5 slopes need 3 ghost layers and are not supported.")
    ific = nfic; ifick = nfic if nk != 2 else 0
    pint[NIJK_XYZ:NIJK_XYZ + 5] = [ni, nj, nk, ific, ifick]
    pint[NIJK:NIJK + 5] = [ni - 1, nj - 1, nk - 1, ific, ifick]
    pint[IJKV] = pint[NIJK] - 2 * ific
    pint[IJKV + 1] = pint[NIJK + 1] - 2 * ific
    pint[IJKV + 2] = pint[NIJK + 2] - 2 * ifick
    pint[NDIMDX] = pint[NIJK] * pint[NIJK + 1] * pint[NIJK + 2] + pint[SHIFTVAR]
    pint[NDIMDX_XYZ] = ni * nj * nk
    typ, nijk_mtr, ndimdx_mtr, neq_ij, neq_k = nature_geom_dom(x, y, zz, ific, ifick)
    pint[NIJK_MTR:NIJK_MTR + 5] = nijk_mtr
    pint[NDIMDX_MTR] = ndimdx_mtr
    pint[ITYPZONE] = typ
    pint[NEQ_IJ] = neq_ij
    pint[NEQ_K] = neq_k
    coe_shift = 1 if pint[ITYPCP] == 2 else 0
    if pint[IFLOW] == 3 and pint[ILES] == 0:
        pint[NEQ_COE] = 6 - coe_shift * 5
    else:
        pint[NEQ_COE] = 5 - coe_shift * 4
    pint[NEQ] = 6 if pint[IFLOW] == 3 else 5
    # fixed mesh: ITYPVENT = 2 (or 3 in 2-D), Fast/FastS/FastS/Metric/allocate_metric.cpp:188-252
    pint[ITYPVENT] = 3 if typ == 3 else 2
    if pint[ITYPVENT] == 2:
        pint[NEQ_VENT] = 3; pint[NDIMDX_VENT] = 1; pint[NIJK_VENT:NIJK_VENT + 5] = 0
    else:
        pint[NEQ_VENT] = 2; pint[NDIMDX_VENT] = ni * nj
        pint[NIJK_VENT:NIJK_VENT + 5] = [1, ni, 0, ific, ifick]
    pint[PT_VOL] = 0
    tz = Internal.getNodeFromName1(o, 'type_zone')
    tz[1][0] = typ

    ss = pint[SIZE_SSDOM:SIZE_SSDOM + 3]
    iv_lu = max(1, pint[IJKV] // ss[0]); jv_lu = max(1, pint[IJKV + 1] // ss[1]); kv_lu = max(1, pint[IJKV + 2] // ss[2])
    nisdom_lu = int(iv_lu * jv_lu * kv_lu)
    ndimdx_dm = 6 * nisdom_lu * nssiter + nssiter + nssiter + 1
    pint[NDIMD_SDM] = ndimdx_dm
    pint[MXSSDOM_LU] = nisdom_lu
    neq = int(pint[NEQ])
    rdm = numpy.zeros(nisdom_lu * neq * 2 * nssiter, numpy.float64)
    it_lu_ssdom = numpy.zeros(nisdom_lu * 5, Internal.E_NpyInt)
    ind_dm = numpy.zeros(ndimdx_dm, Internal.E_NpyInt)
    for nit in range(nssiter):
        nds = 0
        for k in range(kv_lu):
            for j in range(jv_lu):
                for i in range(iv_lu):
                    p = nit * nisdom_lu * 6 + nds * 6
                    ind_dm[p:p + 6] = [1 + ss[0] * i, ss[0] * (i + 1), 1 + ss[1] * j, ss[1] * (j + 1),
                                       1 + ss[2] * k, ss[2] * (k + 1)]
                    if i + 1 == iv_lu: ind_dm[p + 1] = pint[IJKV]
                    if j + 1 == jv_lu: ind_dm[p + 3] = pint[IJKV + 1]
                    if k + 1 == kv_lu: ind_dm[p + 5] = pint[IJKV + 2]
                    nds += 1
    it_lu_ssdom[0:nisdom_lu] = nssiter
    it_lu_ssdom[nisdom_lu:2 * nisdom_lu] = nssiter
    it_lu_ssdom[4 * nisdom_lu:5 * nisdom_lu] = numpy.arange(1, nisdom_lu + 1)

    neq_mtr = 2 * neq_ij + neq_k + 1
    mtr = numpy.zeros(ndimdx_mtr * neq_mtr, numpy.float64)
    if typ == 2:
        mtr[0:4] = cart_metrics(x, y, zz, ific, ifick)
    elif typ == 0:
        mtr[:] = general_metrics(x, y, zz)
    elif typ == 3:
        mtr[:] = planar_metrics(x, y, zz)
    else:
        mtr[:] = homo_metrics(x, y, zz, ific, ifick)
    # collapsed ghost cells: flags dimensioned on the nodes like the reference's (metrics[7]), then the metric
    # correction of init_metric.cpp:318 and, for the Spalart-Allmaras model, that of the wall distance (:324-329)
    deg = degenerate_cells(x, y, zz)
    degen = numpy.zeros((ni, nj, nk), Internal.E_NpyInt, order='F')
    degen[:ni - 1, :nj - 1, :nk - 1] = deg
    tijk_extrap(pint, deg, mtr)
    if pint[IFLOW] == 3:
        sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
        dn = Internal.getNodeFromName1(sol, 'TurbulentDistance') if sol is not None else None
        if dn is not None:
            dist_extrap(pint, deg, dn[1])
    return [mtr, mtr, None, mtr, rdm, ind_dm, it_lu_ssdom, degen.reshape(-1, order='F')]


# ======================================================================================
def getRange(in_bc, pint):
    """CGNS node window of a BC -> interior-cell window + face index 1..6
    (Fast/FastC/FastC/getRange.cpp:57-115)."""
    ific = int(pint[NIJK + 3]); ifick = int(pint[NIJK + 4])
    iv, jv, kv = int(pint[IJKV]), int(pint[IJKV + 1]), int(pint[IJKV + 2])
    f = [0] * 6
    if in_bc[0] == in_bc[1]:
        f[0] = in_bc[0] - 1 if in_bc[0] > 1 else in_bc[0]
        f[0] = max(1, f[0] - 2 * ific); f[1] = f[0]
        f[2] = max(in_bc[2] - ific, 1); f[3] = min(in_bc[3] - ific - 1, jv)
        f[4] = max(in_bc[4] - ifick, 1); f[5] = min(in_bc[5] - ifick - 1, kv)
    elif in_bc[2] == in_bc[3]:
        f[2] = in_bc[2] - 1 if in_bc[2] > 1 else in_bc[2]
        f[2] = max(1, f[2] - 2 * ific); f[3] = f[2]
        f[0] = max(in_bc[0] - ific, 1); f[1] = min(in_bc[1] - ific - 1, iv)
        f[4] = max(in_bc[4] - ifick, 1); f[5] = min(in_bc[5] - ifick - 1, kv)
    elif in_bc[4] == in_bc[5]:
        f[4] = in_bc[4] - 1 if in_bc[4] > 1 else in_bc[4]
        f[4] = max(1, f[4] - 2 * ifick); f[5] = f[4]
        f[0] = max(in_bc[0] - ific, 1); f[1] = min(in_bc[1] - ific - 1, iv)
        f[2] = max(in_bc[2] - ific, 1); f[3] = min(in_bc[3] - ific - 1, jv)
    idir = 0
    if f[0] == f[1] and f[0] == 1: idir = 1
    elif f[0] == f[1] and f[0] == iv: idir = 2
    elif f[2] == f[3] and f[2] == 1: idir = 3
    elif f[2] == f[3] and f[2] == jv: idir = 4
    elif f[4] == f[5] and f[4] == 1: idir = 5
    elif f[4] == f[5] and f[4] == kv: idir = 6
    return f, idir


def _BCcompact(t):
    """Append the BC table to Parameter_int / Parameter_real
    (Fast/FastC/FastC/PyTree.py:2789-3017).  Layout from PT_BC = Parameter_int[70]:
    [nb_bc, ptr_int(nb_bc), ptr_real(nb_bc), {type, idir, imin,imax,jmin,jmax,kmin,kmax,
    nbdata, cumulated sizes(nbdata)} per BC]."""
    for z in Internal.getZones(t):
        o = Internal.getNodeFromName1(z, '.Solver#ownData')
        bcs = Internal.getNodesFromType2(z, 'BC_t')
        nb = len(bcs)
        size_int = 1 + nb * 2 + 9 * nb
        size_real = 0
        for bc in bcs:
            data = _bc_data_arrays(bc)
            size_int += len(data) * len(data)
            for d in data:
                size_real += d[1].size
        pi = Internal.getNodeFromName1(o, 'Parameter_int')
        pr = Internal.getNodeFromName1(o, 'Parameter_real')
        c_int = pi[1].size; c_real = pr[1].size
        datap = numpy.zeros(size_int + c_int, Internal.E_NpyInt)
        datap[0:c_int] = pi[1]
        pi[1] = datap
        if size_real != 0:
            datar = numpy.zeros(size_real + c_real, numpy.float64)
            datar[0:c_real] = pr[1]
            pr[1] = datar
        pint = pi[1]; preal = pr[1]
        pt_bcs = c_int
        pint[PT_BC] = pt_bcs
        pint[pt_bcs] = nb
        i = 1; sz_i = 1 + 2 * nb; sz_r = 0
        for bc in bcs:
            pint[pt_bcs + i] = sz_i + pt_bcs
            pint[pt_bcs + i + nb] = sz_r + c_real
            pt_bc = int(pint[pt_bcs + i])
            btype = Internal.getValue(bc)
            pint[pt_bc] = tagBC(btype)
            rng = Internal.getNodeFromType1(bc, 'IndexRange_t')[1]
            in_bc = [int(rng[0][0]), int(rng[0][1]), int(rng[1][0]), int(rng[1][1]),
                     int(rng[2][0]), int(rng[2][1])]
            fen, idir = getRange(in_bc, pint)
            pint[pt_bc + 1] = idir
            pint[pt_bc + 2:pt_bc + 8] = fen
            data = _bc_data_arrays(bc)
            pint[pt_bc + 8] = len(data)
            ctot = 0; j = 1
            for d in data:
                c = d[1].size
                ctot += c
                pint[pt_bc + 8 + j] = ctot
                deb = c_real + sz_r
                preal[deb:deb + c] = d[1].reshape(-1, order='F')
                d[1] = preal[deb:deb + c]
                sz_r += c; j += 1
            sz_i += 9 + len(data)
            i += 1
    return None


def _bc_data_arrays(bc):
    out = []
    for c in bc[2]:
        for d in c[2]:
            for e in d[2]:
                if e[3] == 'DataArray_t':
                    out.append(e)
    return out


# ======================================================================================
_VARS_P = ['Density', 'VelocityX', 'VelocityY', 'VelocityZ', 'Temperature']


def zone_vars(z):
    o = Internal.getNodeFromName1(z, '.Solver#ownData')
    model = Internal.getValue(Internal.getNodeFromName1(o, 'model'))
    v = list(_VARS_P)
    if model in ('nsspalart', 'NSTurbulent'):
        v.append('TurbulentSANuTilde')
    return v


# Allocator of the compacted level buffers: None = numpy.  fast_amd.PyTree.warmup sets it to page-locked memory of
# the HIP runtime (n float64 -> array, or None where there is none), which the host link copies in place.
LEVEL_ALLOC = None


def _createPrimVars(t):
    """Primitive variables + the _M1/_P1 time levels, then compaction of each level into one
    contiguous buffer with stride NDIMDX (Fast/FastC/FastC/PyTree.py:443-520,2230-2271).
    Returns FIRST_IT (1 when the tree already carried a _M1 level)."""
    first_it = 1
    for b in Internal.getBases(t):
        for z in Internal.getZones(b):
            sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
            if sol is None:
                raise ValueError("createPrimVars: FlowSolution#Centers missing in zone %s." % z[0])
            ref = Internal.getNodeFromName2(z, 'ReferenceState')
            if ref is None:
                ref = Internal.getNodeFromName2(b, 'ReferenceState')
            if ref is None:
                raise ValueError("createPrimVars: ReferenceState is missing in t.")
            adim = _getState(ref)
            gamma = adim[11]; cv = adim[7]
            ro = Internal.getNodeFromName1(sol, 'Density')[1]
            names = {'VelocityX': 'MomentumX', 'VelocityY': 'MomentumY', 'VelocityZ': 'MomentumZ'}
            for v, m in names.items():
                if Internal.getNodeFromName1(sol, v) is None:
                    mom = Internal.getNodeFromName1(sol, m)
                    if mom is None:
                        raise ValueError("createPrimVars: %s or %s required in zone %s." % (v, m, z[0]))
                    Internal.createChild(sol, v, 'DataArray_t', numpy.asfortranarray(mom[1] / ro))
            if Internal.getNodeFromName1(sol, 'Temperature') is None:
                e = Internal.getNodeFromName1(sol, 'EnergyStagnationDensity')
                if e is None:
                    raise ValueError("createPrimVars: Temperature or EnergyStagnationDensity required.")
                u = Internal.getNodeFromName1(sol, 'VelocityX')[1]
                v_ = Internal.getNodeFromName1(sol, 'VelocityY')[1]
                w = Internal.getNodeFromName1(sol, 'VelocityZ')[1]
                T = (e[1] / ro - 0.5 * (u * u + v_ * v_ + w * w)) / cv
                Internal.createChild(sol, 'Temperature', 'DataArray_t', numpy.asfortranarray(T))
            for m in ('MomentumX', 'MomentumY', 'MomentumZ', 'EnergyStagnationDensity',
                      'TurbulentSANuTildeDensity'):
                sol[2] = [c for c in sol[2] if c[0] != m]
            vars_ = zone_vars(z)
            for v in vars_:
                if Internal.getNodeFromName1(sol, v) is None:
                    raise ValueError("createPrimVars: field %s missing in zone %s." % (v, z[0]))
            pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
            ndimdx = int(pint[NDIMDX])
            # selective frequency damping: the filtered state, zero unless the tree carries one
            # (Fast/FastC/FastC/PyTree.py:357-361,678-687)
            levels = ('', '_M1', '_P1') + (('_f',) if int(pint[62]) == 1 else ())
            for suffix in levels:
                buf = LEVEL_ALLOC(len(vars_) * ndimdx) if LEVEL_ALLOC is not None else None
                if buf is None:
                    buf = numpy.empty(len(vars_) * ndimdx, dtype=numpy.float64)
                for c, v in enumerate(vars_):
                    src = Internal.getNodeFromName1(sol, v + suffix)
                    zero = False
                    if src is None:
                        if suffix == '_M1':
                            first_it = 0
                        src = Internal.getNodeFromName1(sol, v)
                        zero = suffix == '_f'
                        node = Internal.createChild(sol, v + suffix, 'DataArray_t', None)
                    else:
                        node = src
                    a = src[1] * 0. if zero else src[1]
                    sh = a.shape
                    view = buf[c * ndimdx:c * ndimdx + a.size]
                    view[:] = a.reshape(-1, order='F')
                    node[1] = view.reshape(sh, order='F')
    return first_it


def switchPointers__(zones, case, order=3):
    """M1 <- current, current <- P1, P1 <- M1 (case 1) by swapping the numpy references in
    the tree (Fast/FastC/FastC/PyTree.py:2292-2315)."""
    if order == 2:
        return None
    for z in zones:
        sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
        for v in zone_vars(z):
            ca = Internal.getNodeFromName1(sol, v)
            caM1 = Internal.getNodeFromName1(sol, v + '_M1')
            caP1 = Internal.getNodeFromName1(sol, v + '_P1')
            if case == 1:
                ta = caM1[1]; caM1[1] = ca[1]; ca[1] = caP1[1]; caP1[1] = ta
            elif case == 2:
                ta = caP1[1]; caP1[1] = ca[1]; ca[1] = caM1[1]; caM1[1] = ta
    return None


def level_buffer(z, suffix=''):
    """The contiguous (NEQ*NDIMDX) buffer behind one time level of a compacted zone."""
    sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
    a = Internal.getNodeFromName1(sol, 'Density' + suffix)[1]
    base = a
    while isinstance(base.base, numpy.ndarray):
        base = base.base
    return base


def fill_task_records(dtloc, zones):
    """Single-thread task records after .Solver#dtloc[dtdim]: nbtask(nssiter), ptiter(nssiter),
    then per sub-iteration one record per zone [nd, nd_subzone, thread_map(1), nthreads,
    topo(3), ind_dm(6)] (record layout: Fast/FastC/FastC/HPC_LAYER/distributeThreads.cpp:520-538,
    consumed at Fast/FastS/FastS/Compute/rhs.cpp:8-12)."""
    nssiter = int(dtloc[0]); sh = int(dtloc[11])
    omp = dtloc[sh:]
    stride = 6 + 7 * 1
    need = 2 * nssiter + nssiter * len(zones) * stride
    if need > omp.size:
        raise ValueError("resize MX_OMP_SIZE_INT: need %d, have %d" % (need, omp.size))
    pos = 2 * nssiter
    for n in range(nssiter):
        omp[n] = len(zones)
        omp[nssiter + n] = pos
        for nd, z in enumerate(zones):
            pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
            omp[pos:pos + stride] = [nd, 0, 0, 1, 1, 1, 1, 1, pint[IJKV], 1, pint[IJKV + 1], 1, pint[IJKV + 2]]
            pos += stride
    return None
