"""Driver for the compiled reference Fortran (oracle/_ref/libfast_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by fast_amd/.

The reference's C++/Python glue cannot be built here (it needs Cassiopee), so this file
restates, for ONE thread and ONE sub-zone per zone, the orchestration that glue performs
around the Fortran entry points:

  * one sub-iteration = navier_stokes_struct (RHS + RK stage) -> joins -> BCs -> corner fix
        Fast/FastS/FastS/Compute/gsdr3.cpp:125-146,251-269,343-349,403-422
        Fast/FastS/FastS/Compute/rhs.cpp:40-58, bcs.cpp:30-41
  * BC dispatch over the table in Parameter_int        Fast/FastS/FastS/BC/BCzone.cpp:193-226,560-568
  * time-level rotation after the last sub-iteration   Fast/FastS/FastS/Compute/computePT.cpp:673-677
  * match-join ghost transfer = indexed copy           Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:440-935

All numerical work is done by the reference's own routines, called through ctypes with
every argument by reference (INTEGER*4 / REAL*8).
"""
import ctypes
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libfast_ref.so")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Open with RTLD_LAZY: the library carries one undefined symbol of the (unused)
    GMRES path, which an eager bind would trip over."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH, mode=os.RTLD_LAZY)
    return _lib


def _i(v):
    return ctypes.byref(ctypes.c_int32(int(v)))


def _d(v):
    return ctypes.byref(ctypes.c_double(float(v)))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# Parameter_int / Parameter_real slots used here (Fast/FastS/FastS/param_solver.h)
NIJK = 0; IJKV = 20; NEQ = 36; NEQ_IJ = 37; NEQ_K = 38; NEQ_COE = 39; NDIMDX = 41
NDIMDX_MTR = 43; ITYPCP = 29; KFLUDOM = 33; PT_BC = 70; ITYPZONE = 25; NSSITER = 58
IFLOW = 27; SLOPE = 64; ILES = 28; SIZE_SSDOM = 30; MXSSDOM_LU = 24; EPSI_NEWTON = 14
DTC = 0; GAMMA = 1; ROINF = 3; PINF = 4; VXINF = 19; VYINF = 20; VZINF = 21; RONUTILDEINF = 8


class RefZone:
    """Host arrays of one zone, laid out as the reference's pyTree + HOOK hold them."""

    def __init__(self, param_int, param_real, x, y, z, ro, mtr, dist=None):
        self.pint = np.ascontiguousarray(param_int, dtype=np.int32).copy()
        self.preal = np.ascontiguousarray(param_real, dtype=np.float64).copy()
        self.x = np.ascontiguousarray(x.reshape(-1, order='F'), dtype=np.float64)
        self.y = np.ascontiguousarray(y.reshape(-1, order='F'), dtype=np.float64)
        self.z = np.ascontiguousarray(z.reshape(-1, order='F'), dtype=np.float64)
        p = self.pint
        self.neq = int(p[NEQ]); self.ndimdx = int(p[NDIMDX])
        n = self.neq * self.ndimdx
        self.ro = np.ascontiguousarray(ro, dtype=np.float64).reshape(-1).copy()
        assert self.ro.size == n
        self.ro_m1 = self.ro.copy()
        self.ro_p1 = self.ro.copy()
        nm = int(p[NDIMDX_MTR]); nij = int(p[NEQ_IJ]); nk = int(p[NEQ_K])
        self.mtr = np.ascontiguousarray(mtr, dtype=np.float64).copy()
        self.ti = self.mtr[0:nm * nij]
        self.tj = self.mtr[nm * nij:2 * nm * nij]
        self.tk = self.mtr[2 * nm * nij:2 * nm * nij + nm * nk]
        self.vol = self.mtr[2 * nm * nij + nm * nk:]
        self.drodm = np.zeros(n)
        self.coe = np.zeros(self.ndimdx * int(p[NEQ_COE]))
        self.xmut = np.zeros(self.ndimdx * 4)
        if dist is not None:     # wall distance of the SA model: second slot of the viscosity array (src_term.for:93-96)
            d = np.ascontiguousarray(dist.reshape(-1, order='F'), dtype=np.float64)
            self.xmut[self.ndimdx:self.ndimdx + d.size] = d
        nwig = 3 * self.ndimdx * (2 if p[SLOPE] in (6, 7) and p[KFLUDOM] == 2 else 1)
        self.wig = np.zeros(nwig)
        self.cfl = np.zeros(3)
        self.iv = int(p[IJKV]); self.jv = int(p[IJKV + 1]); self.kv = int(p[IJKV + 2])
        self.ind_dm = np.array([1, self.iv, 1, self.jv, 1, self.kv], dtype=np.int32)


def ref_metrics(param_int, param_real, x, y, z):
    """ti,tj,tk,vol through the reference's cp_tijk_ / cp_vol_
    (Fast/FastC/FastC/Metric/init_metric.cpp:264-275) for a Cartesian zone
    (loop range (1,1,1,1,1,1), Fast/FastC/FastC/Metric/indice_omp1.h:24-28)."""
    L = lib()
    p = np.ascontiguousarray(param_int, dtype=np.int32)
    nm = int(p[NDIMDX_MTR]); nij = int(p[NEQ_IJ]); nk = int(p[NEQ_K])
    mtr = np.zeros(nm * (2 * nij + nk + 1))
    ti = mtr[0:nm * nij]; tj = mtr[nm * nij:2 * nm * nij]
    tk = mtr[2 * nm * nij:2 * nm * nij + nm * nk]; vol = mtr[2 * nm * nij + nm * nk:]
    xf = np.ascontiguousarray(x.reshape(-1, order='F')); yf = np.ascontiguousarray(y.reshape(-1, order='F'))
    zf = np.ascontiguousarray(z.reshape(-1, order='F'))
    if p[ITYPZONE] == 2:
        ind = np.ones(6, dtype=np.int32)
    else:   # whole zone, ghost cells included (indice_omp1.h:11-16)
        f, fk = int(p[NIJK + 3]), int(p[NIJK + 4])
        ind = np.array([1 - f, int(p[IJKV]) + f, 1 - f, int(p[IJKV + 1]) + f, 1 - fk, int(p[IJKV + 2]) + fk],
                       dtype=np.int32)
        if p[ITYPZONE] == 1:   # k-homogeneous zone: the metrics of the plane k = 1 (indice_omp1.h:19-23)
            ind[4] = ind[5] = 1
    L.cp_tijk_(_p(p), _p(xf), _p(yf), _p(zf), _p(ti), _p(tj), _p(tk), _p(ti), _p(tj), _p(tk), _p(ind))
    L.cp_vol_(_p(p), _p(xf), _p(yf), _p(zf), _p(ti), _p(tj), _p(tk), _p(ti), _p(tj), _p(tk), _p(vol), _p(ind))
    return mtr


def ref_nature_geom_dom(x, y, z, ific, ifick):
    """Zone type through the reference's nature_geom_dom_."""
    L = lib()
    ni, nj, nk = x.shape
    nijk_xyz = np.array([ni, nj, nk, ific, ifick], dtype=np.int32)
    nijk_mtr = np.zeros(5, dtype=np.int32)
    degen = np.zeros(ni * nj * nk, dtype=np.int32)
    ndimdx_mtr = ctypes.c_int32(0); neq_ij = ctypes.c_int32(0); neq_k = ctypes.c_int32(0)
    typ = ctypes.c_int32(0)
    preal = np.zeros(80)
    xf = np.ascontiguousarray(x.reshape(-1, order='F')); yf = np.ascontiguousarray(y.reshape(-1, order='F'))
    zf = np.ascontiguousarray(z.reshape(-1, order='F'))
    L.nature_geom_dom_(_p(nijk_xyz), _i(ni * nj * nk), _p(xf), _p(yf), _p(zf), _p(degen), _i(0), _p(preal),
                       _p(nijk_mtr), ctypes.byref(ndimdx_mtr), ctypes.byref(neq_ij), ctypes.byref(neq_k),
                       ctypes.byref(typ))
    return typ.value, nijk_mtr.tolist(), ndimdx_mtr.value, neq_ij.value, neq_k.value


def ref_degenerate(x, y, z, ific, ifick):
    """Collapsed-cell flags (dimensioned on the nodes) through the reference's nature_geom_dom_."""
    L = lib()
    ni, nj, nk = x.shape
    nijk_xyz = np.array([ni, nj, nk, ific, ifick], dtype=np.int32)
    nijk_mtr = np.zeros(5, dtype=np.int32)
    degen = np.zeros(ni * nj * nk, dtype=np.int32)
    a = ctypes.c_int32(0); b = ctypes.c_int32(0); c = ctypes.c_int32(0); typ = ctypes.c_int32(0)
    preal = np.zeros(80)
    xf = np.ascontiguousarray(x.reshape(-1, order='F')); yf = np.ascontiguousarray(y.reshape(-1, order='F'))
    zf = np.ascontiguousarray(z.reshape(-1, order='F'))
    L.nature_geom_dom_(_p(nijk_xyz), _i(ni * nj * nk), _p(xf), _p(yf), _p(zf), _p(degen), _i(0), _p(preal),
                       _p(nijk_mtr), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(typ))
    return degen


def ref_tijk_extrap(param_int, degen, mtr):
    """tijk_extrap_ of the reference on a metric buffer [ti | tj | tk | vol], with the range of
    Fast/FastC/FastC/Metric/init_metric.cpp:289-322; returns the corrected copy."""
    L = lib()
    p = np.ascontiguousarray(param_int, dtype=np.int32)
    nm = int(p[NDIMDX_MTR]); nij = int(p[NEQ_IJ]); nk = int(p[NEQ_K])
    out = np.array(mtr, dtype=np.float64, copy=True)
    ti = out[0:nm * nij]; tj = out[nm * nij:2 * nm * nij]
    tk = out[2 * nm * nij:2 * nm * nij + nm * nk]; vol = out[2 * nm * nij + nm * nk:]
    ind = np.array([1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2] if p[ITYPZONE] == 0 else 1], dtype=np.int32)
    L.tijk_extrap_(_i(nm), _i(p[41 + 1]), _p(np.ascontiguousarray(p[10:15])), _p(np.ascontiguousarray(p[5:10])),
                   _p(np.ascontiguousarray(p[0:5])), _i(nij), _i(nk), _p(ind), _p(np.ascontiguousarray(degen, dtype=np.int32)),
                   _p(ti), _p(tj), _p(tk), _p(ti), _p(tj), _p(tk), _p(vol))
    return out


def ref_dist_extrap(param_int, degen, dist):
    L = lib()
    p = np.ascontiguousarray(param_int, dtype=np.int32)
    out = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).reshape(-1, order='F')).copy()
    full = np.zeros(int(p[41])); full[:out.size] = out
    ind = np.array([1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]], dtype=np.int32)
    L.dist_extrap_(_i(p[41]), _i(p[42]), _p(np.ascontiguousarray(p[0:5])), _p(np.ascontiguousarray(p[10:15])), _p(ind),
                   _p(np.ascontiguousarray(degen, dtype=np.int32)), _p(full))
    return full[:out.size]


class RefSolver:
    """Runs explicit (RK) time steps of a list of RefZone on the reference Fortran."""

    def __init__(self, zones, joins=None):
        self.zones = zones
        self.joins = joins or []   # list of (rcv_zone, rcv_idx int32[n], dnr_zone, dnr_idx int32[n])
        self.first_it = 0
        # verif: every time step is treated as a verification step (modulo_verif = 1): the residual table of
        # cprdu3s1 is filled at every Newton sub-iteration and the last slot (residual only) is run
        self.verif = False
        self.mx_synchro = 1000
        self.lok = np.zeros(self.mx_synchro * 4, dtype=np.int32)
        self.timer = np.zeros(64)
        self.stage1_rhs = None
        # OpenMP threads of the RHS + update (oracle/ref_mt.c drives the reference's own thread decomposition and locks)
        self.threads = 1
        # 'implicit_local' (lssiter_loc = 1): Newton sub-iterations on the sub-zones that have not converged; kmod =
        # modulo_verif, the period of the verification steps that re-evaluate them
        self.local = False
        self.kmod = 1
        self._sz_ready = False

    # -- one sub-iteration of one zone: RHS + update -------------------------------------
    def _navier_stokes(self, nd, zn, nitrun, nstep, ro_ssiter):
        L = lib()
        p = zn.pint
        nssiter = int(p[NSSITER])
        ijkv_sdm = np.zeros(3, dtype=np.int32)
        ind_dm_socket = np.zeros(12, dtype=np.int32)
        topo_socket = np.ones(3, dtype=np.int32)
        topo_omp = np.ones(3, dtype=np.int32)
        lmin = 4 if p[ITYPCP] == 2 else 10
        L.indice_boucle_lu_(_i(nd), _i(1), _i(1), _i(lmin), _p(zn.ind_dm), _p(topo_socket), _p(ind_dm_socket))
        temps = (nitrun - 1) * zn.preal[DTC]
        self.lok[:] = 0
        norm_kry = ctypes.c_double(0.)
        alias = zn.ro
        if self.threads > 1:
            from . import oracle as _O
            n = self.threads
            if getattr(self, '_mt', None) is None or self._mt[0] != n:
                self._mt = (n, np.zeros(self.mx_synchro * n, dtype=np.int32), np.zeros(64 * n, dtype=np.int32),
                            np.zeros(2 * n), np.zeros(3 * n))
            _, lok, work, timer, cfl = self._mt
            f = _O.lib().fo_ref_mt_subiter
            f.restype = ctypes.c_int
            vp = lambda a: ctypes.c_void_p(a.ctypes.data)
            f(ctypes.cast(L.navier_stokes_struct_, ctypes.c_void_p), ctypes.cast(L.indice_boucle_lu_, ctypes.c_void_p),
              ctypes.c_int(n), ctypes.c_int(nd), ctypes.c_int(nstep), ctypes.c_int(nssiter), ctypes.c_int(nitrun),
              ctypes.c_int(self.first_it), vp(p), vp(zn.preal), ctypes.c_double(temps), vp(zn.ind_dm), vp(lok),
              ctypes.c_int(self.mx_synchro), vp(work), vp(timer), vp(cfl), vp(alias), vp(zn.x), vp(zn.y), vp(zn.z),
              vp(zn.ro), vp(zn.ro_m1), vp(zn.ro_p1), vp(ro_ssiter), vp(zn.xmut), vp(zn.ti), vp(zn.tj), vp(zn.tk),
              vp(zn.vol), vp(zn.wig), vp(zn.drodm), vp(zn.coe))
            return
        L.navier_stokes_struct_(
            _i(nd), _i(1), _i(1), _i(1), _i(0), _i(1), _i(1), _i(1), _i(self.mx_synchro),
            _i(0), _i(0), _i(1), _i(nstep), _i(nssiter), _i(nitrun), _i(self.first_it),
            _i(0), _i(0),
            _p(p), _p(zn.preal), _d(temps),
            _p(ijkv_sdm), _p(zn.ind_dm), _p(ind_dm_socket), _p(zn.ind_dm), _p(topo_socket),
            _p(self.lok), _p(topo_omp), _p(self.timer),
            _p(alias), ctypes.byref(norm_kry), _p(zn.cfl),
            _p(zn.x), _p(zn.y), _p(zn.z), _p(alias), _p(alias),
            _p(zn.ro), _p(zn.ro_m1), _p(zn.ro_p1), _p(ro_ssiter),
            _p(zn.xmut),
            _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.vol),
            _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.vol),
            _p(alias), _p(alias), _p(alias),
            _p(zn.wig), _p(zn.wig), _p(zn.wig),
            _p(zn.drodm), _p(zn.coe), _p(alias), _p(alias), _p(alias))

    # -- BCs of one zone on array `rop` ------------------------------------------------------
    def _bczone(self, nd, zn, nstep, rop):
        L = lib()
        p = zn.pint
        pt_bcs = int(p[PT_BC]); nb_bc = int(p[pt_bcs])
        state = np.zeros(6)
        r = zn.preal
        state[0] = r[ROINF]; state[1] = r[VXINF] * r[ROINF]; state[2] = r[VYINF] * r[ROINF]
        state[3] = r[VZINF] * r[ROINF]
        state[4] = r[PINF] / (r[GAMMA] - 1.) + 0.5 * (r[VXINF] * state[1] + r[VYINF] * state[2] + r[VZINF] * state[3])
        state[5] = r[RONUTILDEINF]
        ind_cl = np.zeros(6, dtype=np.int32); ind_cl119 = np.zeros(6, dtype=np.int32)
        lskip = ctypes.c_int32(0)
        for ndf in range(nb_bc):
            pt_bc = int(p[pt_bcs + 1 + ndf])
            idir = int(p[pt_bc + 1]); btype = int(p[pt_bc])
            fen = np.ascontiguousarray(p[pt_bc + 2:pt_bc + 8])
            if idir <= 2: tijk, neq_mtr = zn.ti, int(p[NEQ_IJ])
            elif idir <= 4: tijk, neq_mtr = zn.tj, int(p[NEQ_IJ])
            else: tijk, neq_mtr = zn.tk, int(p[NEQ_K])
            L.indice_cl_sdm_(_i(idir), _i(0), ctypes.byref(lskip), _i(btype), _i(p[NIJK + 3]), _i(p[NIJK + 4]),
                             _p(np.ascontiguousarray(p[IJKV:IJKV + 3])), _p(fen), _p(zn.ind_dm), _p(zn.ind_dm),
                             _p(ind_cl), _p(ind_cl119))
            if lskip.value != 0:
                continue
            if btype == 11:
                L.bvbs_periodique_(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl119),
                                   _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                                   _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state))
            elif btype == 0:
                L.bvbs_extrapolate_(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl119),
                                    _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                                    _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state))
            elif btype == 14:
                L.bvbs_extrapolate_ransles_(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl119),
                                            _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                                            _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state))
            elif btype == 22:
                pass   # BCOverlap: ghost cells come from the join transfer
            elif btype in (16, 32):
                # Fast/FastS/FastS/BC/BCzone.cpp:539-558,570-592 with INCREMENT_BC.h: one array of window data
                nbdata = int(p[pt_bc + 8])
                if nbdata < 1 or nbdata > 2 or (btype == 32 and p[IFLOW] <= 1):
                    raise NotImplementedError("ref_harness: BC tag %d with %d data arrays" % (btype, nbdata))
                off = int(p[pt_bcs + 1 + ndf + nb_bc])
                size0 = int(p[pt_bc + 9])
                if size0 == 1:
                    off += 1; size0 = int(p[pt_bc + 10])
                data = zn.preal[off:]
                if idir <= 2: inc_bc = [fen[3] - fen[2] + 1, fen[2], fen[4]]
                elif idir <= 4: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[4]]
                else: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[2]]
                inc_bc = np.array(inc_bc, dtype=np.int32)
                fn = L.bvbs_outpres_ if btype == 16 else L.bvbs_wall_viscous_isothermal_
                fn(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl),
                   _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                   _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state),
                   _p(data), _i(int(p[pt_bc + 9])), _p(inc_bc))
            elif btype == 21:
                # Fast/FastS/FastS/BC/BCzone.cpp:400-415: mass flow rate per unit area, one value per face cell
                if int(p[pt_bc + 8]) != 1:
                    raise NotImplementedError("ref_harness: BCOutMFR with %d data arrays" % int(p[pt_bc + 8]))
                off = int(p[pt_bcs + 1 + ndf + nb_bc])
                if idir <= 2: inc_bc = [fen[3] - fen[2] + 1, fen[2], fen[4]]
                elif idir <= 4: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[4]]
                else: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[2]]
                inc_bc = np.array(inc_bc, dtype=np.int32)
                L.bvbs_outmfr_(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl),
                               _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                               _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state),
                               _p(zn.preal[off:]), _i(int(p[pt_bc + 9])), _p(inc_bc))
            elif btype in (17, 20):
                # Fast/FastS/FastS/BC/BCzone.cpp:594-618 (BCInj1: direction, stagnation pressure and enthalpy per face
                # cell) and :377-398 (BCInjMFR: direction, mass flow rate per unit area and stagnation enthalpy)
                nbdata = int(p[pt_bc + 8]); size0 = int(p[pt_bc + 9])
                if nbdata != int(p[NEQ]):
                    raise NotImplementedError("ref_harness: BCInj1 / BCInjMFR with %d data arrays" % nbdata)
                off = int(p[pt_bcs + 1 + ndf + nb_bc])
                if idir <= 2: inc_bc = [fen[3] - fen[2] + 1, fen[2], fen[4]]
                elif idir <= 4: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[4]]
                else: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[2]]
                inc_bc = np.array(inc_bc, dtype=np.int32)
                d = [zn.preal[off + e * size0:] for e in range(6)]
                fn = L.bvbs_inflow_newton_ if btype == 17 else L.bvbs_injmfr_
                fn(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl),
                                      _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                                      _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut),
                                      _p(state), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), _p(d[5]),
                                      _i(size0), _p(inc_bc))
            elif btype == 2 and int(p[pt_bc + 8]) != 0 and int(p[pt_bc + 9]) > 6:
                # Fast/FastS/FastS/BC/BCzone.cpp:417-448: conservative state per face cell, 5 or 6 arrays of one size
                nbdata = int(p[pt_bc + 8]); size0 = int(p[pt_bc + 9])
                if size0 <= 6 or nbdata != int(p[NEQ]):
                    raise NotImplementedError("ref_harness: BCInflowSupersonic with %d data arrays of %d" % (nbdata, size0))
                off = int(p[pt_bcs + 1 + ndf + nb_bc])
                if idir <= 2: inc_bc = [fen[3] - fen[2] + 1, fen[2], fen[4]]
                elif idir <= 4: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[4]]
                else: inc_bc = [fen[1] - fen[0] + 1, fen[0], fen[2]]
                inc_bc = np.array(inc_bc, dtype=np.int32)
                d = [zn.preal[off + e * size0:] for e in range(6)]
                L.bvbs_inflow_supersonic_fich_(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl),
                                               _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                                               _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut),
                                               _p(state), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), _p(d[5]),
                                               _i(size0), _p(inc_bc))
            elif btype in (1, 2, 3, 4, 5, 6, 10, 13):
                # dispatch of Fast/FastS/FastS/BC/BCzone.cpp:279-283,299-310,417-425,451-461,513-523 (these use ind_CL)
                st = state
                nbdata = int(p[pt_bc + 8])
                if nbdata != 0:     # DEFAULT_STATE, BCzone.cpp:40-48: the BC's own uniform state
                    dd = zn.preal[int(p[pt_bcs + 1 + ndf + nb_bc]):]
                    if btype in (1, 2, 10, 13) and nbdata == 21:
                        st = np.array([dd[3], dd[4], dd[5], dd[6], dd[7], dd[16]])
                    elif btype in (1, 2, 10, 13) and nbdata == 1 and int(p[pt_bc + 9]) == 6:
                        st = np.ascontiguousarray(dd[:6])
                    else:
                        raise NotImplementedError("ref_harness: BC data")
                state_save, state = state, st
                if btype == 1:
                    fn = L.bvbs_farfield_
                elif btype == 13:
                    fn = L.bvbs_inflow_
                elif btype == 2:
                    fn = L.bvbs_inflow_supersonic_
                elif btype == 10:
                    fn = L.bvbs_outflow_
                elif btype == 3 or (btype == 4 and p[IFLOW] == 1) or btype == 5:
                    fn = L.bvbs_wall_inviscid_
                elif p[IFLOW] > 1:
                    fn = L.bvbs_wall_viscous_adia_
                else:
                    raise NotImplementedError("ref_harness: BC tag %d with IFLOW %d" % (btype, p[IFLOW]))
                fn(_i(idir), _i(0), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(ind_cl),
                   _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
                   _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(rop), _p(zn.xmut), _p(state))
                state = state_save
            else:
                raise NotImplementedError("ref_harness: BC tag %d" % btype)
        L.correct_coins_(_i(nd), _p(p), _p(zn.ind_dm), _p(rop))

    def _transfers(self, arrays):
        """type-0 ('direct') match joins: rcv[rcvPts] = dnr[donorPts] for every variable."""
        for (zr, ridx, zd, didx) in self.joins:
            nr = self.zones[zr].ndimdx; ndd = self.zones[zd].ndimdx
            for v in range(self.zones[zr].neq):
                arrays[zr][v * nr + ridx] = arrays[zd][v * ndd + didx]

    def fill_ghostcells(self, which='ro'):
        arrays = [getattr(z, which) for z in self.zones]
        self._transfers(arrays)
        for nd, zn in enumerate(self.zones):
            self._bczone(nd, zn, 1, arrays[nd])

    # -- implicit: LU-SGS solve + Newton update of one zone ---------------------------------
    def _bczone_rhs(self, nd, zn, nstep):
        """BCzone with lrhs = 1 on the Newton residual (Fast/FastS/FastS/Compute/lhs.cpp:53-68,
        Fast/FastS/FastS/BC/BCzone.cpp:93-191,193-226,451-523): on every zone face that carries a wall-type
        BC the residual is first extrapolated into the ghost cells over the whole face, then every BC of
        that face is applied to it (walls as walls, the others as extrapolation); the LU range grows by
        one ghost layer on those faces; edges and corners are then averaged (correct_coins).
        Returns the LU range."""
        L = lib()
        p = zn.pint
        pt_bcs = int(p[PT_BC]); nb_bc = int(p[pt_bcs])
        lskip = [1] * 6
        itest = [1, zn.iv, 1, zn.jv, 1, zn.kv]
        for ndf in range(nb_bc):
            pt_bc = int(p[pt_bcs + 1 + ndf]); bt = int(p[pt_bc]); ipara = int(p[pt_bc + 1]) - 1
            if (3 <= bt <= 7) or bt in (12, 18, 30):
                if bt in (12, 18, 30) or (bt == 4 and p[IFLOW] == 1 and False):
                    raise NotImplementedError("ref_harness: implicit treatment of BC tag %d" % bt)
                if zn.ind_dm[ipara] == itest[ipara]:      # BCzone.cpp:101: the range of the thread reaches the face
                    lskip[ipara] = 0
        shift = zn.ind_dm.copy()
        for d in range(3):
            shift[2 * d] -= (1 - lskip[2 * d]); shift[2 * d + 1] += (1 - lskip[2 * d + 1])
        if all(lskip):
            return shift
        state = np.zeros(6)
        ind_cl = np.zeros(6, dtype=np.int32); ind_cl119 = np.zeros(6, dtype=np.int32)
        lsk = ctypes.c_int32(0)
        ijkv = np.ascontiguousarray(p[IJKV:IJKV + 3])

        def call(fn, idir, cl, tijk, neq_mtr):
            fn(_i(idir), _i(1), _i(nstep), _i(neq_mtr), _d(1.), _p(p), _p(cl),
               _p(zn.preal), _d(5. / 6.), _d(2. / 6.), _d(-1. / 6.),
               _p(zn.x), _p(zn.y), _p(zn.z), _p(zn.ro), _p(tijk), _p(zn.drodm), _p(zn.xmut), _p(state))

        for idir0 in range(6):
            if lskip[idir0]:
                continue
            ind_rhs = zn.ind_dm.copy()
            d = idir0 // 2
            if idir0 % 2 == 0: ind_rhs[2 * d] = 1; ind_rhs[2 * d + 1] = 1
            else: ind_rhs[2 * d] = ind_rhs[2 * d + 1]
            L.indice_cl_sdm_(_i(idir0 + 1), _i(0), ctypes.byref(lsk), _i(0), _i(p[NIJK + 3]), _i(p[NIJK + 4]),
                             _p(ijkv), _p(ind_rhs), _p(zn.ind_dm), _p(zn.ind_dm), _p(ind_cl), _p(ind_cl119))
            if lsk.value == 0:
                call(L.bvbs_extrapolate_, idir0 + 1, ind_cl119, zn.ti, int(p[NEQ_IJ]))
        for ndf in range(nb_bc):
            pt_bc = int(p[pt_bcs + 1 + ndf]); bt = int(p[pt_bc]); idir = int(p[pt_bc + 1])
            if lskip[idir - 1]:
                continue
            if idir <= 2: tijk, neq_mtr = zn.ti, int(p[NEQ_IJ])
            elif idir <= 4: tijk, neq_mtr = zn.tj, int(p[NEQ_IJ])
            else: tijk, neq_mtr = zn.tk, int(p[NEQ_K])
            fen = np.ascontiguousarray(p[pt_bc + 2:pt_bc + 8])
            L.indice_cl_sdm_(_i(idir), _i(0), ctypes.byref(lsk), _i(bt), _i(p[NIJK + 3]), _i(p[NIJK + 4]),
                             _p(ijkv), _p(fen), _p(zn.ind_dm), _p(zn.ind_dm), _p(ind_cl), _p(ind_cl119))
            if lsk.value != 0:
                continue
            if bt in (0, 1, 2, 10, 11, 14, 16, 17, 20, 21):
                call(L.bvbs_extrapolate_, idir, ind_cl119, tijk, neq_mtr)
            elif bt == 3 or (bt == 4 and p[IFLOW] == 1) or bt == 5:
                call(L.bvbs_wall_inviscid_, idir, ind_cl, tijk, neq_mtr)
            elif bt in (4, 6) and p[IFLOW] > 1:
                call(L.bvbs_wall_viscous_adia_, idir, ind_cl, tijk, neq_mtr)
            elif bt == 22:
                pass
            else:
                raise NotImplementedError("ref_harness: implicit treatment of BC tag %d" % bt)
        L.correct_coins_(_i(nd), _p(p), _p(shift), _p(zn.drodm))
        return shift

    def lu_thread_blocks(self, nd, zn, nthreads):
        """ind_dm_thread of the nthreads OpenMP threads in zone nd with omp_mode 0 (every zone cut among all the threads,
        Fast/FastC/FastC/HPC_LAYER/distributeThreads.cpp:90-95): indice_boucle_lu_ of the compiled reference; threads
        left without cells are dropped."""
        L = lib()
        topo = np.ones(3, dtype=np.int32)
        blocks = []
        for ith in range(1, int(nthreads) + 1):
            ind = np.zeros(6, dtype=np.int32)
            L.indice_boucle_lu_(_i(nd), _i(ith), _i(int(nthreads)), _i(10), _p(zn.ind_dm), _p(topo), _p(ind))
            if ind[1] >= ind[0] and ind[3] >= ind[2] and ind[5] >= ind[4]:
                blocks.append(ind)
        return np.array(blocks, dtype=np.int32)

    def _invlu(self, nd, zn, nitrun, nstep, ro_ssiter):
        """Fast/FastS/FastS/Compute/lhs.cpp:135-161.  nb_relax > 1 or lu_match = 1: the increments are built in a work
        array (iptdrodm_out = ssortmp, lhs.cpp:148) next to the accumulator ssor, both dimensioned on the window of
        the thread, here the whole zone with its ghost cells (Compute/ssor.cpp, lhs.cpp:184-200)."""
        L = lib()
        p = zn.pint
        # lu_match = 1: the copy of the residual into the ghost cells of joins (lhs.cpp:1-15) is left out: with one
        # thread per zone no sweep reads them (BC/BCzone.cpp:122-149)
        shift = self._bczone_rhs(nd, zn, nstep)
        if p[71] > 1 and p[75] == 0 and getattr(self, 'lu_threads', 1) > 1:
            # every thread relaxes its own block on work arrays of its own window (Compute/ssor.cpp:52-70: the block
            # with its ghost layers; LU/prep_lussor.h)
            neq = int(p[36]); f = int(p[NIJK + 3]); fk = int(p[NIJK + 4])
            if getattr(zn, 'ssor_thread', None) is None:
                zn.ssor_thread = {}
            for ib, blk in enumerate(self.lu_thread_blocks(nd, zn, self.lu_threads)):
                sh = blk.copy()
                for q in range(6):
                    if blk[q] == zn.ind_dm[q]:
                        sh[q] = shift[q]
                size = int((blk[1] - blk[0] + 1 + 2 * f) * (blk[3] - blk[2] + 1 + 2 * f) * (blk[5] - blk[4] + 1 + 2 * fk))
                if ib not in zn.ssor_thread:
                    zn.ssor_thread[ib] = (np.zeros(size * neq), np.zeros(size * neq))
                ssor, ssortmp = zn.ssor_thread[ib]
                L.invlu_(_i(nd), _i(nstep), _i(nitrun), _p(p), _p(zn.preal), _p(sh), _p(blk), _i(1),
                         _p(zn.ro_p1), _p(ro_ssiter), _p(zn.drodm), _p(ssortmp),
                         _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.ro), _p(zn.ro), _p(zn.ro),
                         _p(zn.coe), _p(ssor), _i(size))
            return
        if p[71] > 1 or p[75] != 0:
            ssor_size = int(p[NIJK]) * int(p[NIJK + 1]) * int(p[NIJK + 2])
            neq = int(p[36])
            if getattr(zn, 'ssor', None) is None:
                zn.ssor = np.zeros(ssor_size * neq); zn.ssortmp = np.zeros(ssor_size * neq)
            L.invlu_(_i(nd), _i(nstep), _i(nitrun), _p(p), _p(zn.preal), _p(shift), _p(zn.ind_dm), _i(1),
                     _p(zn.ro_p1), _p(ro_ssiter), _p(zn.drodm), _p(zn.ssortmp),
                     _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.ro), _p(zn.ro), _p(zn.ro),
                     _p(zn.coe), _p(zn.ssor), _i(ssor_size))
            return
        dummy = np.zeros(8)
        if getattr(self, 'lu_threads', 1) > 1:
            # every thread solves its own block (lhs.cpp:129-161); its LU range takes the ghost layer of a wall face of
            # the zone where the block touches it (BC/BCzone.cpp:99-127).  The blocks do not read each other's cells,
            # so one call after the other gives what the threads give.
            for blk in self.lu_thread_blocks(nd, zn, self.lu_threads):
                sh = blk.copy()
                for q in range(6):
                    if blk[q] == zn.ind_dm[q]:
                        sh[q] = shift[q]
                L.invlu_(_i(nd), _i(nstep), _i(nitrun), _p(p), _p(zn.preal), _p(sh), _p(blk), _i(1),
                         _p(zn.ro_p1), _p(ro_ssiter), _p(zn.drodm), _p(zn.drodm),
                         _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.ro), _p(zn.ro), _p(zn.ro),
                         _p(zn.coe), _p(dummy), _i(1))
            return
        L.invlu_(_i(nd), _i(nstep), _i(nitrun), _p(p), _p(zn.preal), _p(shift), _p(zn.ind_dm), _i(1),
                 _p(zn.ro_p1), _p(ro_ssiter), _p(zn.drodm), _p(zn.drodm),
                 _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.ro), _p(zn.ro), _p(zn.ro),
                 _p(zn.coe), _p(dummy), _i(1))

    def _cprdu3s1(self, nd, zn, nitrun, nstep):
        """Residual table of one zone at sub-iteration nstep of a verification step: cprdu3s1_ as called by
        Fast/FastS/FastS/Compute/residus_navier.h:12-43 (one thread, every sub-block computed: no_lu and
        nisdom_residu as Fast/FastC/FastC/Compute/init_ssiter_bloc.for fills them without 'implicit_local'; it_bloc
        reset at the first sub-iteration and incremented before each, init_ssiter_bloc.for + Compute/ssor.cpp:14-20)."""
        L = lib()
        p = zn.pint
        nssiter = int(p[NSSITER]); neq = zn.neq
        ijkv = np.ascontiguousarray(p[IJKV:IJKV + 3])
        ss = np.ascontiguousarray(p[SIZE_SSDOM:SIZE_SSDOM + 3])
        ijkv_lu = np.array([max(1, int(ijkv[d]) // int(ss[d])) for d in range(3)], dtype=np.int32)
        ndim_rdm = int(ijkv_lu[0]) * int(ijkv_lu[1]) * int(ijkv_lu[2])
        mx = max(int(p[MXSSDOM_LU]), ndim_rdm)
        if nstep == 1:
            zn.rdm = np.zeros(nssiter * neq * 2 * ndim_rdm)
            zn.it_lu = [np.zeros(mx, dtype=np.int32) for _ in range(4)]
            zn.no_lu = np.zeros(mx, dtype=np.int32); zn.no_lu[:ndim_rdm] = np.arange(1, ndim_rdm + 1)
            zn.nisdom_residu = np.full(nssiter, ndim_rdm, dtype=np.int32)
            zn.it_bloc = 0
        zn.it_bloc += 1
        L.cprdu3s1_(_i(nd), _i(nstep), _i(nssiter), _i(neq), _i(zn.ndimdx), _i(ndim_rdm), _i(nitrun),
                    _i(p[IFLOW]), _i(p[ILES]), _i(1), _i(1), _i(0),
                    _i(mx), _i(zn.it_bloc), _d(zn.preal[EPSI_NEWTON]),
                    _p(ss), _p(zn.nisdom_residu), _p(np.ascontiguousarray(p[NIJK:NIJK + 5])), _p(ijkv), _p(ijkv_lu),
                    _p(zn.ind_dm), _p(zn.it_lu[0]), _p(zn.it_lu[1]), _p(zn.it_lu[2]), _p(zn.it_lu[3]), _p(zn.no_lu),
                    _p(zn.drodm), _p(zn.rdm))

    def residual_history(self):
        """Per zone (L2, Loo)[sub-iteration][equation] of the last verification step, assembled from the table of
        cprdu3s1_ as Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:112-146 does."""
        from .oracle import assemble_rdm
        return [assemble_rdm(zn.pint, zn.preal, zn.rdm, int(zn.pint[NSSITER]), zn.it_bloc) for zn in self.zones]

    def display_reference(self, nitrun, names, history, lft=1, iverb=0, record=True):
        """dpssiter_ and conv2pytree_ of the compiled reference for every zone, driven as
        Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:107-175 drives them after a verification step: the norms
        assembled from the cprdu3s1 table go to dpssiter_ as rdm(it_bloc, neq, 2), which prints the table of the
        sub-iterations and fills cvg(4 neq); conv2pytree_ appends the record to the zone's convergence history.
        history[nd] = dict(last=int32[1], itnum=int32[nrec], l2, oo, l2_diff, oo_diff = float64[neq * nrec]), the
        arrays of the ZoneConvergenceHistory node (Fast/FastS/FastS/PyTree.py:1955-1990), updated in place.
        record = False: dpssiter_ only (nothing is appended).  Returns what the reference printed (a string)."""
        import tempfile
        L = lib()
        L._FortranAioBeginFlush.restype = ctypes.c_void_p
        L._FortranAioBeginFlush.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L._FortranAioEndIoStatement.argtypes = [ctypes.c_void_p]

        def flush6():
            L._FortranAioEndIoStatement(L._FortranAioBeginFlush(6, None, 0))

        from .oracle import assemble_rdm
        with tempfile.TemporaryFile() as tmp:
            flush6()
            saved = os.dup(1)
            os.dup2(tmp.fileno(), 1)
            try:
                for nd, zn in enumerate(self.zones):
                    neq, nssiter = zn.neq, int(zn.pint[NSSITER])
                    l2, loo = assemble_rdm(zn.pint, zn.preal, zn.rdm, nssiter, zn.it_bloc)
                    glob = np.zeros(2 * neq * zn.it_bloc)
                    for ne in range(neq):       # display_ss_iteration.cpp:137-141
                        glob[zn.it_bloc * ne:zn.it_bloc * (ne + 1)] = l2[:, ne]
                        glob[zn.it_bloc * (ne + neq):zn.it_bloc * (ne + neq + 1)] = loo[:, ne]
                    cvg = np.zeros(4 * neq)
                    name = names[nd].encode()
                    L.dpssiter_(_i(nitrun), _i(neq), _i(zn.it_bloc), _i(nssiter), _p(zn.pint), _i(lft), _i(iverb),
                                ctypes.c_char_p(name), _i(len(name)), _p(glob), _p(cvg))
                    if not record:
                        continue
                    h = history[nd]
                    L.conv2pytree_(_p(cvg), _i(nitrun), _i(neq), _p(h['last']), ctypes.c_char_p(name), _i(len(name)),
                                   _i(lft), _i(h['itnum'].size), _i(nd), _p(h['itnum']), _p(h['l2']), _p(h['oo']),
                                   _p(h['l2_diff']), _p(h['oo_diff']))
                flush6()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            tmp.seek(0)
            return tmp.read().decode()

    def tgv_sums(self):
        """Per zone [enstrophy sum, kinetic-energy sum, volume] of the current state: post_ of the compiled reference
        (POST/post.for: velocity gradient of cp_gradu with the 4-point face average c1 = 7/12, c2 = 1/12, then
        cptaylor.for), called as Fast/FastS/FastS/POST/computePT_enstrophy.cpp:176-204 does for one thread."""
        L = lib()
        out = []
        for nd, zn in enumerate(self.zones):
            p = zn.pint
            ijkv_sdm = np.zeros(3, dtype=np.int32)
            ind_dm_socket = np.zeros(12, dtype=np.int32)
            topo_socket = np.ones(3, dtype=np.int32)
            topo_omp = np.ones(3, dtype=np.int32)
            L.indice_boucle_lu_(_i(nd), _i(1), _i(1), _i(4), _p(zn.ind_dm), _p(topo_socket), _p(ind_dm_socket))
            self.lok[:] = 0
            grad = np.zeros(9 * zn.ndimdx)
            tke = ctypes.c_double(0.); enst = ctypes.c_double(0.); cpt = ctypes.c_double(0.)
            L.post_(_i(nd), _i(1), _i(1), _i(1), _i(1), _i(self.mx_synchro), _i(3),
                    _p(p), _p(zn.preal), ctypes.byref(tke), ctypes.byref(enst), ctypes.byref(cpt),
                    _p(ijkv_sdm), _p(zn.ind_dm), _p(ind_dm_socket), _p(zn.ind_dm), _p(topo_omp), _p(topo_socket),
                    _p(self.lok), _p(zn.ro), _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.vol), _p(grad))
            out.append(np.array([enst.value, tke.value, cpt.value]))
        return out

    def _sfd(self, zn, nitrun, level=None):
        """sfd_ on the interior of the level the sub-iteration has written (lhs.cpp:184)."""
        if int(zn.pint[62]) != 1:
            return
        if getattr(zn, 'rof', None) is None:
            zn.rof = np.zeros_like(zn.ro)
        lib().sfd_(_p(zn.pint), _p(zn.preal), _i(nitrun), _p(zn.ind_dm), _p(zn.ro_p1 if level is None else level), _p(zn.rof),
                   _p(zn.coe), _p(zn.vol))

    def step_implicit(self, nitrun):
        """One implicit time step: nssiter-1 Newton sub-iterations (the last slot of nssiter is the
        verification pass, skipped when lssiter_verif = 0: Fast/FastS/FastS/Compute/computePT.cpp:579-585),
        each = Newton residual (navier_stokes_struct: fluxes + core3as2), LU-SGS + update into P1
        (invlu), joins and BCs on P1 (gsdr3.cpp:137-140); then the level rotation of
        Fast/FastS/FastS/PyTree.py:231-232."""
        z0 = self.zones[0]
        nssiter = int(z0.pint[NSSITER])
        assert z0.pint[ITYPCP] == 1
        for nstep in range(1, nssiter + (1 if self.verif else 0)):
            for nd, zn in enumerate(self.zones):
                ssiter = zn.ro if nstep == 1 else zn.ro_p1
                self._navier_stokes(nd, zn, nitrun, nstep, ssiter)
                if nstep == 1:
                    zn.stage1_rhs = zn.drodm.copy()
                if self.verif:     # lhs.cpp:90: the residual of the interior, before the LU solve
                    self._cprdu3s1(nd, zn, nitrun, nstep)
            # selective frequency damping, once per step, in the last slot that runs (lhs.cpp:181-185): after the
            # LU solve of slot nssiter-1, or, on a verification step, after the residual of slot nssiter
            sfd_now = nstep == (nssiter if self.verif else nssiter - 1)
            if nstep == nssiter:   # verification slot: lexit_lu = 1 (souszones_list.cpp:83), nothing is advanced
                for nd, zn in enumerate(self.zones):
                    self._sfd(zn, nitrun)
                break
            for nd, zn in enumerate(self.zones):
                ssiter = zn.ro if nstep == 1 else zn.ro_p1
                self._invlu(nd, zn, nitrun, nstep, ssiter)
                if sfd_now:
                    self._sfd(zn, nitrun)
            cl = [zn.ro_p1 for zn in self.zones]
            self._transfers(cl)
            for nd, zn in enumerate(self.zones):
                self._bczone(nd, zn, nstep, cl[nd])
        self.first_it = 1
        for zn in self.zones:
            zn.ro_m1, zn.ro, zn.ro_p1 = zn.ro, zn.ro_p1, zn.ro_m1

    # -- 'implicit_local': Newton sub-iterations on sub-zones ----------------------------------------------------
    def _sz_setup(self):
        """ind_dm / it_lu_ssdom arrays of every zone, sized and filled as
        Fast/FastS/FastS/Metric/allocate_metric.cpp:398-471 does, the table of skip_lu, and the calls of
        souszones_list that warmup makes (Fast/FastS/FastS/PyTree.py:391-394: nitrun = 1, every sub-iteration, both
        passes because no list exists yet)."""
        z0 = self.zones[0]
        nssiter = int(z0.pint[NSSITER])
        self.iskip_lu = np.zeros(2 * nssiter, dtype=np.int32)
        self.dtloc = np.zeros(16, dtype=np.int32)
        for zn in self.zones:
            p = zn.pint
            ijkv = p[IJKV:IJKV + 3]; ss = p[SIZE_SSDOM:SIZE_SSDOM + 3]
            zn.ijkv_lu = np.array([max(1, int(ijkv[d]) // int(ss[d])) for d in range(3)], dtype=np.int32)
            n = int(np.prod(zn.ijkv_lu)); mx = int(p[MXSSDOM_LU])
            assert mx >= n
            zn.sz_n = n; zn.sz_mx = mx
            zn.ind_dm_tab = np.zeros(6 * mx * nssiter + 2 * nssiter + 1, dtype=np.int32)
            zn.it_tab = np.zeros(5 * mx, dtype=np.int32)
            zn.it_tab[0:2 * mx] = nssiter
            zn.it_tab[4 * mx:5 * mx] = np.arange(1, mx + 1)
            zn.rdm = np.zeros(nssiter * zn.neq * 2 * n)
        for nstep in range(1, nssiter + 1):
            self._souszones_c(1, nstep, 1)
            self._souszones_c(0, nstep, 0)
        self._sz_ready = True

    def _sz_views(self, zn):
        nssiter = int(zn.pint[NSSITER]); mx = zn.sz_mx
        t = zn.ind_dm_tab; q = 6 * mx * nssiter
        return dict(ind_dm=t[0:q], nisdom_residu=t[q:q + nssiter], nidom_loc=t[q + nssiter:q + 2 * nssiter],
                    it_bloc=t[q + 2 * nssiter:], it_lu=zn.it_tab[0:mx], it_target=zn.it_tab[mx:2 * mx],
                    it_target_old=zn.it_tab[2 * mx:3 * mx], it_temp=zn.it_tab[3 * mx:4 * mx], no_lu=zn.it_tab[4 * mx:])

    def _souszones_c(self, nitrun, nstep, flag_res):
        """souszones_list_c for every zone (Fast/FastC/FastC/Compute/souszones_list.cpp:45-79): init_ssiter_bloc_ of the
        compiled reference on the arrays of the zone."""
        L = lib()
        for nd, zn in enumerate(self.zones):
            p = zn.pint
            v = self._sz_views(zn)
            nssiter = int(p[NSSITER])
            ijk_lu = np.zeros(3 * zn.sz_mx, dtype=np.int32)
            L.init_ssiter_bloc_(_i(nd), _i(nstep), _i(nssiter), _i(nitrun), _i(1), _i(p[ITYPCP]), _i(flag_res),
                                _p(np.ascontiguousarray(p[IJKV:IJKV + 3])), _p(zn.ijkv_lu), _p(ijk_lu),
                                _p(np.ascontiguousarray(p[SIZE_SSDOM:SIZE_SSDOM + 3])), _i(zn.sz_mx), _p(self.iskip_lu),
                                _p(self.dtloc), _p(v['ind_dm']), _p(v['nidom_loc']), _p(v['it_bloc']),
                                _p(v['nisdom_residu']), _p(v['it_lu']), _p(v['it_target']), _p(v['it_target_old']),
                                _p(v['no_lu']), _p(p))

    def _souszones(self, nitrun, nstep):
        """souszones_list as Fast/FastS/FastS/PyTree.py:122 calls it before every sub-iteration
        (Fast/FastC/FastC/Compute/souszones_list.cpp:161-196 with lssiter_loc = 1): returns (lexit_lu, lssiter_verif)."""
        nssiter = int(self.zones[0].pint[NSSITER])
        kmod = self.kmod
        lexit_lu = lssiter_verif = 0
        init_exit = 0
        if nitrun % kmod == 0:
            init_exit = 1
            self._souszones_c(nitrun, nstep, 1)
            lssiter_verif = 1
            lexit_lu = 1 if nstep == nssiter else 0
        if (nitrun - 1) % kmod == 0:
            self._souszones_c(nitrun - 1, nstep, 0)
            lssiter_verif = 1
            lexit_lu = 1 if nstep == nssiter else 0
            if init_exit == 0:
                lexit_lu = lssiter_verif = 0
                init_exit = 2
            if kmod == 1:
                lssiter_verif = 1
                lexit_lu = 1 if nstep == nssiter else 0
        if init_exit == 0:
            lexit_lu = lssiter_verif = 0
        return lexit_lu, lssiter_verif

    def sub_zones(self, nd, nstep):
        """Boxes (i1, i2, j1, j2, k1, k2) of zone nd at sub-iteration nstep."""
        zn = self.zones[nd]
        v = self._sz_views(zn)
        nb = int(v['nidom_loc'][nstep - 1])
        q = (nstep - 1) * 6 * zn.sz_mx
        return [v['ind_dm'][q + 6 * s:q + 6 * s + 6].copy() for s in range(nb)]

    def step_implicit_local(self, nitrun):
        """One time step of 'implicit_local': as step_implicit, every zone cut at every sub-iteration into the boxes of
        init_ssiter_bloc / ssdom_lu_ijk (one task per box: Fast/FastS/FastS/Compute/rhs.cpp:9-58, lhs.cpp:18-161,
        bcs.cpp:14-41), the residual of cprdu3s1 per block of SIZE_SSDOM cells feeding the targets of the next
        evaluation.  The joins move every point whatever the boxes (gsdr3.cpp:343-349)."""
        L = lib()
        if not self._sz_ready:
            self._sz_setup()
        z0 = self.zones[0]
        nssiter = int(z0.pint[NSSITER])
        assert z0.pint[ITYPCP] == 1
        self.last_boxes = []
        for nstep in range(1, nssiter + 1):
            lexit_lu, verif = self._souszones(nitrun, nstep)
            if verif == 0 and nstep == nssiter:
                break
            boxes = [self.sub_zones(nd, nstep) for nd in range(len(self.zones))]
            self.last_boxes.append(boxes)
            whole = [zn.ind_dm for zn in self.zones]
            shifts = [[] for _ in self.zones]
            try:
                for nd, zn in enumerate(self.zones):
                    ssiter = zn.ro if nstep == 1 else zn.ro_p1
                    for box in boxes[nd]:
                        zn.ind_dm = box
                        self._navier_stokes(nd, zn, nitrun, nstep, ssiter)
                    if nstep == 1:
                        zn.stage1_rhs = zn.drodm.copy()
                for nd, zn in enumerate(self.zones):
                    p = zn.pint
                    v = self._sz_views(zn)
                    for s, box in enumerate(boxes[nd]):
                        zn.ind_dm = box
                        shifts[nd].append(self._bczone_rhs(nd, zn, nstep))
                        if verif and s == 0:
                            v['it_bloc'][0] += 1      # Compute/ssor.cpp:14-20
                            zn.it_bloc = int(v['it_bloc'][0])
                            L.cprdu3s1_(_i(nd), _i(nstep), _i(nssiter), _i(zn.neq), _i(zn.ndimdx), _i(zn.sz_n), _i(nitrun),
                                        _i(p[IFLOW]), _i(p[ILES]), _i(1), _i(1), _i(0),
                                        _i(zn.sz_mx), _i(zn.it_bloc), _d(zn.preal[EPSI_NEWTON]),
                                        _p(np.ascontiguousarray(p[SIZE_SSDOM:SIZE_SSDOM + 3])), _p(v['nisdom_residu']),
                                        _p(np.ascontiguousarray(p[NIJK:NIJK + 5])),
                                        _p(np.ascontiguousarray(p[IJKV:IJKV + 3])), _p(zn.ijkv_lu), _p(box),
                                        _p(v['it_lu']), _p(v['it_target']), _p(v['it_target_old']), _p(v['it_temp']),
                                        _p(v['no_lu']), _p(zn.drodm), _p(zn.rdm))
                sfd_now = nstep == (nssiter if verif else nssiter - 1)
                if lexit_lu:
                    for nd, zn in enumerate(self.zones):
                        for box in boxes[nd]:
                            zn.ind_dm = box
                            self._sfd(zn, nitrun)
                    continue
                for nd, zn in enumerate(self.zones):
                    ssiter = zn.ro if nstep == 1 else zn.ro_p1
                    dummy = np.zeros(8)
                    for box, shift in zip(boxes[nd], shifts[nd]):
                        L.invlu_(_i(nd), _i(nstep), _i(nitrun), _p(zn.pint), _p(zn.preal), _p(shift), _p(box), _i(1),
                                 _p(zn.ro_p1), _p(ssiter), _p(zn.drodm), _p(zn.drodm),
                                 _p(zn.ti), _p(zn.tj), _p(zn.tk), _p(zn.ro), _p(zn.ro), _p(zn.ro),
                                 _p(zn.coe), _p(dummy), _i(1))
                        if sfd_now:
                            zn.ind_dm = box
                            self._sfd(zn, nitrun)
                cl = [zn.ro_p1 for zn in self.zones]
                self._transfers(cl)
                for nd, zn in enumerate(self.zones):
                    for box in boxes[nd]:
                        zn.ind_dm = box
                        self._bczone(nd, zn, nstep, cl[nd])
            finally:
                for zn, w in zip(self.zones, whole):
                    zn.ind_dm = w
        self.first_it = 1
        for zn in self.zones:
            zn.ro_m1, zn.ro, zn.ro_p1 = zn.ro, zn.ro_p1, zn.ro_m1

    def step(self, nitrun):
        """One time step = nssiter sub-iterations, then M1 <- N <- P1 rotation."""
        z0 = self.zones[0]
        nssiter = int(z0.pint[NSSITER])
        if z0.pint[ITYPCP] == 1:
            return self.step_implicit_local(nitrun) if self.local else self.step_implicit(nitrun)
        assert z0.pint[ITYPCP] == 2, "RefSolver.step: explicit and implicit (LU-SGS) schemes only"
        for nstep in range(1, nssiter + 1):
            cl = []
            for nd, zn in enumerate(self.zones):
                if nstep == 1: ssiter, c = zn.ro, zn.ro_p1
                elif nstep % 2 == 0: ssiter, c = zn.ro_p1, zn.ro
                else: ssiter, c = zn.ro, zn.ro_p1
                if nstep == 1:
                    zn.preal[22] = 0. if nd == 0 else zn.preal[22]   # gsdr3.cpp:155
                self._navier_stokes(nd, zn, nitrun, nstep, ssiter)
                if nstep == 1:
                    zn.stage1_rhs = zn.drodm.copy()
                if nstep == nssiter - 1:      # lhs.cpp:181-185: the explicit loop runs the same condition (no verification slot)
                    self._sfd(zn, nitrun, c)
                cl.append(c)
            self._transfers(cl)
            for nd, zn in enumerate(self.zones):
                self._bczone(nd, zn, nstep, cl[nd])
        self.first_it = 1
        if nssiter % 2 == 1:   # Fast/FastS/FastS/PyTree.py:233 (order = nitmax%2+2)
            for zn in self.zones:
                zn.ro_m1, zn.ro, zn.ro_p1 = zn.ro, zn.ro_p1, zn.ro_m1

    def residuals(self):
        """Per zone, per equation: sqrt(sum (coe*drodm)^2 / ncell)/dt and max|coe*drodm|/dt of
        the first-stage balance, over interior cells (the norm of
        Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:128-141 applied to the explicit
        stage-1 increment; the reference only tabulates it for implicit zones)."""
        out = []
        for zn in self.zones:
            p = zn.pint
            ni, nj, nk = int(p[NIJK]), int(p[NIJK + 1]), int(p[NIJK + 2])
            f = int(p[NIJK + 3]); fk = int(p[NIJK + 4])
            d = zn.stage1_rhs.reshape(zn.neq, -1)[:, :ni * nj * nk].reshape(zn.neq, nk, nj, ni)
            d = d[:, fk:nk - fk, f:nj - f, f:ni - f]
            c = zn.coe[:zn.ndimdx][:ni * nj * nk].reshape(nk, nj, ni)[fk:nk - fk, f:nj - f, f:ni - f]
            # implicit: stage1_rhs is the Newton residual, already scaled by dt/vol (core3as2.for)
            r = d * c[None] if p[ITYPCP] == 2 else d
            n = r[0].size
            dt = zn.preal[DTC]
            out.append((np.sqrt((r * r).reshape(zn.neq, -1).sum(axis=1) / n) / dt,
                        np.abs(r).reshape(zn.neq, -1).max(axis=1) / dt))
        return out
