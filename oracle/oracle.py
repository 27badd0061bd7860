"""ctypes wrapper of the CPU restatement (oracle/fasts_oracle.c).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker; never imported by fast_amd/.
"""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfasts_oracle.so")
_lib = None

NIJK = 0; IJKV = 20; ITYPZONE = 25; NEQ = 36; NEQ_IJ = 37; NEQ_K = 38; NDIMDX = 41; NDIMDX_MTR = 43
NSSITER = 58; DTC = 0; ITYPCP = 29; NEQ_COE = 39; IFLOW = 27; SIZE_SSDOM = 30


def residual_blocks(pint):
    """Sub-blocks per direction of the residual table (Fast/FastS/FastS/Compute/residus_navier.h:12-16)."""
    return [max(1, int(pint[IJKV + d]) // int(pint[SIZE_SSDOM + d])) for d in range(3)]


def assemble_rdm(pint, preal, rdm, nssiter, it_bloc):
    """L2 and Loo norms per Newton sub-iteration from the per-sub-block table rdm(nssiter, neq, 2, nblk) (Fortran
    order, as cprdu3s1 fills it), the loop of Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:112-146: sums of
    squares added over the sub-blocks k_lu, j_lu, i_lu (i fastest), L2 = sqrt(sum / ncell) / dt with
    1 / ncell evaluated in single precision (:98), Loo = largest magnitude / dt.  Returns (l2, loo)[it_bloc][neq]."""
    neq = int(pint[NEQ])
    lu = residual_blocks(pint)
    nblk = lu[0] * lu[1] * lu[2]
    r = np.asarray(rdm).reshape(nblk, 2, neq, nssiter)      # C view of rdm(nssiter, neq, 2, nblk)
    xpt_1 = 1. / float(np.float32(int(pint[IJKV]) * int(pint[IJKV + 1]) * int(pint[IJKV + 2])))
    dt = float(preal[DTC])
    l2 = np.zeros((it_bloc, neq)); loo = np.zeros((it_bloc, neq))
    for ne in range(neq):
        for i in range(it_bloc):
            rmax = 0.; rmoy = 0.
            for b in range(nblk):
                rmax = max(rmax, float(r[b, 1, ne, i]))
                rmoy = rmoy + float(r[b, 0, ne, i])
            l2[i, ne] = np.sqrt(rmoy * xpt_1) / dt
            loo[i, ne] = rmax / dt
    return l2, loo


def build():
    src = os.path.join(HERE, "fasts_oracle.c")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libfasts_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.fo_step.restype = ctypes.c_int
        _lib.fo_step_mt.restype = ctypes.c_int
        _lib.fo_bczone.restype = ctypes.c_int
        _lib.fo_subiter_implicit.restype = ctypes.c_int
        _lib.fo_subiter_implicit_blocks.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleZone:
    """One stand-alone zone (all boundaries local) advanced by the C restatement."""

    def __init__(self, param_int, param_real, ro, mtr, dist=None):
        self.pint = np.ascontiguousarray(param_int, dtype=np.int32).copy()
        self.preal = np.ascontiguousarray(param_real, dtype=np.float64).copy()
        p = self.pint
        self.neq = int(p[NEQ]); self.ndimdx = int(p[NDIMDX])
        n = self.neq * self.ndimdx
        self.ro = np.ascontiguousarray(ro, dtype=np.float64).reshape(-1).copy()
        self.ro_m1 = self.ro.copy(); self.ro_p1 = self.ro.copy()
        self.mtr = np.ascontiguousarray(mtr, dtype=np.float64).copy()
        if int(p[ITYPZONE]) == 1:
            # k-homogeneous zone (extruded mesh): ti, tj(:,1..2), tk(:,1) and vol of one plane, no stride along k.
            # The reference runs its *_3dhomo_* fragments, which are the general ones without the terms that the
            # missing components would multiply; here the vectors are padded with zeros and the zone goes through the
            # general restatement with the same zero stride (adding a product by zero changes no bit of a sum), the z
            # derivatives of the viscous flux and of the SA source excepted, which the C code takes in the factored
            # form of those fragments.  tests/test_oracle_vs_ref.py holds this bit-exact against the compiled fragments.
            nm = int(p[NDIMDX_MTR])
            full = np.zeros(10 * nm)
            full[0:2 * nm] = self.mtr[0:2 * nm]; full[3 * nm:5 * nm] = self.mtr[2 * nm:4 * nm]
            full[8 * nm:9 * nm] = self.mtr[4 * nm:5 * nm]; full[9 * nm:] = self.mtr[5 * nm:6 * nm]
            self.mtr = full
            p[NEQ_IJ] = 3; p[NEQ_K] = 3
        nm = int(p[NDIMDX_MTR]); nij = int(p[NEQ_IJ]); nk = int(p[NEQ_K])
        self.ti = self.mtr[0:nm * nij]; self.tj = self.mtr[nm * nij:2 * nm * nij]
        self.tk = self.mtr[2 * nm * nij:2 * nm * nij + nm * nk]
        self.vol = self.mtr[2 * nm * nij + nm * nk:]
        self.implicit = int(p[ITYPCP]) == 1
        self.first_it = 0          # FIRST_IT of the reference: 0 until one time step is done (BDF1)
        self.drodm = np.zeros(n); self.coe = np.zeros(self.ndimdx * max(1, int(p[NEQ_COE])))
        # viscosity; the SA model keeps the wall distance in a second slot (src_term.for:93-96)
        self.xmut = np.zeros(2 * self.ndimdx)
        # wiggle sensor (3 arrays), then with the o3sc slope the shock sensor (3 more when the scheme is senseur)
        self.wig = np.zeros((6 if (int(p[64]) == 6 and int(p[33]) == 2) else 3) * self.ndimdx)
        if dist is not None:
            d = np.ascontiguousarray(np.asarray(dist).reshape(-1, order='F'), dtype=np.float64)
            self.xmut[self.ndimdx:self.ndimdx + d.size] = d
        self.stage1_rhs = np.zeros(n)
        # residual table of a verification step: rdm(nssiter, neq, 2, nblk) and the number of slots filled
        self.verif = False
        self.rdm = None; self.it_bloc = 0
        # OpenMP threads of the explicit sub-iteration (k-slabs; same result bit for bit)
        self.threads = 1
        # selective frequency damping: filtered state (zero at the start) and the number of the time step
        self.rof = None; self.nitrun = 0

    def fill_ghostcells(self):
        if lib().fo_bczone(_p(self.pint), _p(self.preal), _p(self.ro), _p(self.ti), _p(self.tj), _p(self.tk)) != 0:
            raise NotImplementedError("oracle: unsupported BC tag")

    def nsub(self):
        """Sub-iterations that advance the solution: the last slot of an implicit step is the
        verification pass (Fast/FastS/FastS/Compute/computePT.cpp:579-585)."""
        n = int(self.pint[NSSITER])
        return n - 1 if self.implicit else n

    def sfd(self, level=None):
        """Selective frequency damping of the level just written (once per time step)."""
        if int(self.pint[62]) != 1:
            return
        if self.rof is None:
            self.rof = np.zeros_like(self.ro)
        lib().fo_sfd(_p(self.pint), _p(self.preal), ctypes.c_int(self.nitrun), _p(self.ro_p1 if level is None else level),
                     _p(self.rof), _p(self.coe), _p(self.vol))

    def sfd_stage(self):
        """Sub-iteration after which sfd runs on a step without a verification slot (lhs.cpp:181-185: nitcfg = nssiter - 1;
        the explicit loop runs the same condition, so an explicit step is filtered after its last stage but one)."""
        return int(self.pint[NSSITER]) - 1

    def step(self):
        self.nitrun += 1
        if self.implicit:
            for nstep in range(1, self.nsub() + 1):
                self.stage(nstep)
                if nstep == self.nsub() and not self.verif:
                    self.sfd()      # before the joins and BCs of the slot (lhs.cpp:181-185)
                self.bcs(nstep)
            self.verification_slot()
            return self.rotate()
        if int(self.pint[62]) == 1:     # filtered between two stages: the split form
            for nstep in range(1, int(self.pint[NSSITER]) + 1):
                self.stage(nstep)
                if nstep == self.sfd_stage():
                    self.sfd(self.roles(nstep)[1])
                self.bcs(nstep)
            return self.rotate()
        rc = lib().fo_step_mt(_p(self.pint), _p(self.preal), _p(self.ro), _p(self.ro_p1), _p(self.drodm),
                              _p(self.coe), _p(self.xmut), _p(self.wig), _p(self.ti), _p(self.tj),
                              _p(self.tk), _p(self.vol), _p(self.stage1_rhs), ctypes.c_int(self.threads))
        if rc != 0:
            raise NotImplementedError("oracle: unsupported BC tag")
        if int(self.pint[NSSITER]) % 2 == 1:
            self.ro_m1, self.ro, self.ro_p1 = self.ro, self.ro_p1, self.ro_m1

    def residuals(self):
        """L2 and Loo norms of the first sub-iteration: explicit = dt/vol * balance, implicit = Newton
        residual (display_ss_iteration.cpp:128-141), both divided by dt."""
        p = self.pint
        if self.implicit:
            ni, nj, nk = int(p[NIJK]), int(p[NIJK + 1]), int(p[NIJK + 2]); f = int(p[NIJK + 3]); fk = int(p[NIJK + 4])
            d = self.stage1_rhs.reshape(self.neq, -1)[:, :ni * nj * nk].reshape(self.neq, nk, nj, ni)
            d = d[:, fk:nk - fk, f:nj - f, f:ni - f].reshape(self.neq, -1)
            dt = self.preal[DTC]
            return np.sqrt((d * d).sum(axis=1) / d.shape[1]) / dt, np.abs(d).max(axis=1) / dt
        s = np.zeros(5); m = np.zeros(5)
        lib().fo_residual(_p(self.pint), _p(self.stage1_rhs), _p(self.coe), _p(s), _p(m))
        p = self.pint
        n = int(p[IJKV]) * int(p[IJKV + 1]) * int(p[IJKV + 2])
        dt = self.preal[DTC]
        return np.sqrt(s / n) / dt, m / dt

    def tgv_sums(self):
        """[enstrophy sum, kinetic-energy sum, volume] of the current state (POST/post.for, cptaylor.for)."""
        out = np.zeros(3)
        if lib().fo_tgv_sums(_p(self.pint), _p(self.ro), _p(self.ti), _p(self.tj), _p(self.tk), _p(self.vol), _p(out)) != 0:
            raise NotImplementedError("oracle: kinetic energy / enstrophy on Cartesian zones only")
        return out

    def cfl(self):
        """[max, min, mean] of the local CFL number of the current state: what the first sub-iteration of a
        verification step stores in CFL_minmaxmoy (cptst3.for:114-120, computePT.cpp:519,627-643)."""
        p = self.pint
        ind = np.array([1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]], dtype=np.int32)
        if int(p[IFLOW]) == 3:
            lib().fo_vispalart(_p(p), _p(self.preal), _p(ind), _p(self.ro), _p(self.xmut))
        elif int(p[IFLOW]) == 2:
            lib().fo_invist(_p(p), _p(self.preal), _p(ind), _p(self.ro), _p(self.xmut))
        out = np.array([0., 100000000., 0.])
        lib().fo_cpcfl3(_p(p), _p(self.preal), _p(self.ro), _p(self.xmut), _p(self.ti), _p(self.tj), _p(self.tk),
                        _p(self.vol), _p(out))
        out[2] /= int(p[IJKV]) * int(p[IJKV + 1]) * int(p[IJKV + 2])
        return out

    # -- split form of a time step, for multi-zone jobs -----------------------------------
    def roles(self, nstep):
        """(input, output) arrays of sub-iteration nstep (Fast/FastS/FastS/Compute/gsdr3.cpp:137-145)."""
        if self.implicit:      # Newton iterate -> P1 (gsdr3.cpp:137-140)
            return (self.ro if nstep == 1 else self.ro_p1, self.ro_p1)
        return (self.ro, self.ro_p1) if nstep % 2 == 1 else (self.ro_p1, self.ro)

    def _record_residual(self, nstep, rhs):
        """cprdu3s1 on a verification step: slot it_bloc = nstep of the residual table."""
        nssiter = int(self.pint[NSSITER]); neq = self.neq
        lu = residual_blocks(self.pint); nblk = lu[0] * lu[1] * lu[2]
        if nstep == 1:
            self.rdm = np.zeros((nblk, 2, neq, nssiter))
        sums = np.zeros((nblk, neq)); maxs = np.zeros((nblk, neq))
        lib().fo_cprdu3s1(_p(self.pint), _p(rhs), _p(sums), _p(maxs))
        self.rdm[:, 0, :, nstep - 1] = sums
        self.rdm[:, 1, :, nstep - 1] = maxs
        self.it_bloc = nstep

    def residual_history(self):
        """(L2, Loo)[sub-iteration][equation] of the last verification step."""
        return assemble_rdm(self.pint, self.preal, self.rdm, int(self.pint[NSSITER]), self.it_bloc)

    def verification_slot(self):
        """Last of the NSSITER slots of an implicit step: run only on verification steps, residual only."""
        if self.implicit and self.verif:
            self.stage(int(self.pint[NSSITER]))
            self.sfd()              # no join or BC follows in this slot (gsdr3.cpp:343,417)

    def stage(self, nstep):
        if self.implicit:
            rhs_only = nstep == int(self.pint[NSSITER])
            keep = self.stage1_rhs if nstep == 1 else (self.drodm_keep() if self.verif else None)
            # lu_blocks: (n, 6) int32 ranges of interior cells, the blocks of the reference's OpenMP threads, or None
            blk = getattr(self, 'lu_blocks', None)
            nblk = 0 if blk is None else len(blk)
            rc = lib().fo_subiter_implicit_blocks(_p(self.pint), _p(self.preal), ctypes.c_int(nstep), ctypes.c_int(self.first_it),
                                                  _p(self.ro), _p(self.ro_m1), _p(self.ro_p1), _p(self.drodm), _p(self.coe),
                                                  _p(self.xmut), _p(self.wig), _p(self.ti), _p(self.tj), _p(self.tk), _p(self.vol),
                                                  _p(keep) if keep is not None else None, ctypes.c_int(int(rhs_only)),
                                                  ctypes.c_int(nblk), _p(blk) if nblk else None)
            if rc != 0:
                raise NotImplementedError("oracle: implicit options outside LU-SGS nb_relax=1 / lu_match=0 / no wall")
            if self.verif:
                self._record_residual(nstep, keep)
            return
        rin, rout = self.roles(nstep)
        lib().fo_subiter_mt(_p(self.pint), _p(self.preal), ctypes.c_int(nstep), _p(rin), _p(rout),
                            _p(self.drodm), _p(self.coe), _p(self.xmut), _p(self.wig), _p(self.ti),
                            _p(self.tj), _p(self.tk), _p(self.vol), ctypes.c_int(self.threads))
        if nstep == 1:
            self.stage1_rhs[:] = self.drodm

    def drodm_keep(self):
        if getattr(self, '_keep', None) is None:
            self._keep = np.zeros(self.neq * self.ndimdx)
        return self._keep

    def bcs(self, nstep):
        if lib().fo_bczone(_p(self.pint), _p(self.preal), _p(self.roles(nstep)[1]), _p(self.ti), _p(self.tj),
                           _p(self.tk)) != 0:
            raise NotImplementedError("oracle: unsupported BC tag")

    def rotate(self):
        # explicit: the result sits in P1 only after an odd number of stages; implicit: always
        # (Fast/FastS/FastS/PyTree.py:231-233)
        if self.implicit or int(self.pint[NSSITER]) % 2 == 1:
            self.ro_m1, self.ro, self.ro_p1 = self.ro, self.ro_p1, self.ro_m1
        self.first_it = 1


class OracleSolver:
    """Several zones coupled by match joins: per sub-iteration, every zone is advanced, then the
    ghost cells of the joins are copied (receiver[rcv_pts] = donor[donor_pts] for every variable,
    Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:440-935, 'direct' type), then BCs and the
    edge/corner fix (order of Fast/FastS/FastS/Compute/gsdr3.cpp:251-269,343-349,403-422).
    joins = [(rcv_zone_index, rcv_pts, donor_zone_index, donor_pts)]."""

    def __init__(self, zones, joins=None):
        self.zones = zones
        self.joins = joins or []

    def transfers(self, arrays):
        for (zr, ridx, zd, didx) in self.joins:
            nr = self.zones[zr].ndimdx; ndd = self.zones[zd].ndimdx
            for v in range(self.zones[zr].neq):
                arrays[zr][v * nr + ridx] = arrays[zd][v * ndd + didx]

    def fill_ghostcells(self):
        self.transfers([z.ro for z in self.zones])
        for z in self.zones:
            z.fill_ghostcells()

    def step(self):
        for z in self.zones:
            z.nitrun += 1
        for nstep in range(1, self.zones[0].nsub() + 1):
            for z in self.zones:
                z.stage(nstep)
                if z.implicit and nstep == z.nsub() and not z.verif:
                    z.sfd()
                if not z.implicit and nstep == z.sfd_stage():
                    z.sfd(z.roles(nstep)[1])
            self.transfers([z.roles(nstep)[1] for z in self.zones])
            for z in self.zones:
                z.bcs(nstep)
        for z in self.zones:
            z.verification_slot()
        for z in self.zones:
            z.rotate()
