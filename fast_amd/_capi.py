"""ctypes binding of libfasts_hip.so (C ABI: include/fasts_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, FastsHipError is
raised with the library's message.
"""
import ctypes
import os
import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FASTS_HIP_LIB", os.path.join(HERE, "csrc", "libfasts_hip.so"))

FS_LEVEL_N, FS_LEVEL_M1, FS_LEVEL_P1 = 1, 2, 4
FS_LEVEL_FILTERED = 8      # fs_download only: the filtered state of selective frequency damping

SYMBOLS = [
    "fs_version", "fs_create", "fs_destroy", "fs_last_error", "fs_set_stream", "fs_set_verif", "fs_set_first_it",
    "fs_zone_add", "fs_zone_set_distance", "fs_zone_set_filtered", "fs_zone_set_lu_blocks", "fs_set_lu_threads", "fs_zone_lu_blocks", "fs_join_add", "fs_send_add", "fs_upload", "fs_download", "fs_compute",
    "fs_stage", "fs_stage_part", "fs_set_comm_stream", "fs_halo_count", "fs_halo_pack", "fs_halo_unpack", "fs_local_joins", "fs_bcs",
    "fs_rotate", "fs_fill_ghostcells", "fs_apply_bc", "fs_residuals", "fs_residual_history", "fs_cfl", "fs_tgv_sums", "fs_tgv_sums_static", "fs_sync",
    "fs_stage_kernel_ms", "fs_debug_fetch",
    "fs_comm_unique_id", "fs_comm_init", "fs_zone_set_global_id", "fs_send_add_remote", "fs_comm_setup", "fs_halo_exchange",
    "fs_comm_allreduce", "fs_joins_set", "fs_stream_triad", "fs_host_alloc", "fs_host_free",
    "fs_subzone_skip_lu", "fs_subzone_boxes", "fs_subzone_residual", "fs_set_local_newton", "fs_subzone_list",
    "fs_subzone_targets",
]

COMM_ID_BYTES = 128


class FastsHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FastsHipError(
                "fast_amd: native library %s is missing; build it with "
                "`make -C fast_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.fs_version.restype = ctypes.c_char_p
        L.fs_last_error.restype = ctypes.c_char_p
        L.fs_last_error.argtypes = [ctypes.c_void_p]
        L.fs_destroy.restype = None
        L.fs_destroy.argtypes = [ctypes.c_void_p]
        for name in SYMBOLS:
            f = getattr(L, name)
            if name not in ("fs_version", "fs_last_error", "fs_destroy"):
                f.restype = ctypes.c_int
        _lib = L
    return _lib


def pinned_empty(n, device=0):
    """n float64 of page-locked host memory as a numpy array (fs_host_alloc), released when the array and its views
    are gone; None where the runtime has none to give (no GPU: the caller allocates ordinary memory).
    FASTS_HIP_PIN=0: never."""
    import weakref
    import numpy
    if os.environ.get('FASTS_HIP_PIN', '1') == '0' or n <= 0:
        return None
    try:
        L = lib()
    except (FastsHipError, OSError):
        return None
    p = ctypes.c_void_p()
    L.fs_host_alloc.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    L.fs_host_free.argtypes = [ctypes.c_void_p]
    if L.fs_host_alloc(ctypes.c_int(int(device)), ctypes.byref(p), ctypes.c_size_t(8 * int(n))) != 0 or not p.value:
        return None
    raw = (ctypes.c_double * int(n)).from_address(p.value)
    weakref.finalize(raw, L.fs_host_free, ctypes.c_void_p(p.value))
    return numpy.frombuffer(raw, dtype=numpy.float64)


def _pi(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _pd(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def comm_unique_id():
    """Identifier of a new communicator (bytes); one rank makes it, every rank passes it to Context.comm_init."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    rc = lib().fs_comm_unique_id(buf, ctypes.c_int(COMM_ID_BYTES))
    if rc != 0:
        raise FastsHipError("fs_comm_unique_id failed (%d)" % rc)
    return buf.raw


class Context:
    """One solver context = one GPU."""

    def __init__(self, device=0):
        L = lib()
        self._h = ctypes.c_void_p()
        rc = L.fs_create(ctypes.byref(self._h), ctypes.c_int(device))
        if rc != 0:
            raise FastsHipError("fs_create failed (%d): %s" % (rc, L.fs_last_error(None).decode()))
        self._keep = []   # host arrays the library borrows

    def _chk(self, rc, what):
        if rc < 0:
            raise FastsHipError("%s failed (%d): %s" % (what, rc, lib().fs_last_error(self._h).decode()))
        return rc

    def close(self):
        if self._h:
            lib().fs_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._chk(lib().fs_set_stream(self._h, ctypes.c_void_p(stream_ptr)), "fs_set_stream")

    def set_verif(self, modulo_verif):
        self._chk(lib().fs_set_verif(self._h, ctypes.c_int(int(modulo_verif))), "fs_set_verif")

    def set_local_newton(self, lssiter_loc):
        """1: Newton sub-iterations on the sub-zones that have not converged ('implicit_local'); before upload()."""
        self._chk(lib().fs_set_local_newton(self._h, ctypes.c_int(int(lssiter_loc))), "fs_set_local_newton")

    def subzone_list(self, zone, nstep, max_boxes=4096):
        """Boxes (i1, i2, j1, j2, k1, k2) zone `zone` works on at sub-iteration nstep: int32[nbox, 6]."""
        out = numpy.zeros(6 * max_boxes, dtype=numpy.int32)
        n = self._chk(lib().fs_subzone_list(self._h, ctypes.c_int(int(zone)), ctypes.c_int(int(nstep)),
                                            out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(max_boxes)), "fs_subzone_list")
        return out[:6 * n].reshape(n, 6)

    def subzone_targets(self, zone, max_blocks=1 << 20):
        """Sub-iterations the blocks of zone `zone` are given at the next evaluation of the sub-zones."""
        out = numpy.zeros(max_blocks, dtype=numpy.int32)
        n = self._chk(lib().fs_subzone_targets(self._h, ctypes.c_int(int(zone)), out.ctypes.data_as(ctypes.c_void_p),
                                               ctypes.c_int(max_blocks)), "fs_subzone_targets")
        return out[:n].copy()

    def zone_set_distance(self, zone, dist):
        d = numpy.ascontiguousarray(numpy.asarray(dist).reshape(-1, order='F'), dtype=numpy.float64)
        self._keep.append(d)
        self._chk(lib().fs_zone_set_distance(self._h, ctypes.c_int(int(zone)), _pd(d), ctypes.c_int(d.size)),
                  "fs_zone_set_distance")

    def zone_set_filtered(self, zone, rof):
        """rof: the contiguous NEQ * NDIMDX buffer of the filtered state (kept by the caller; refreshed by
        download(FS_LEVEL_FILTERED))."""
        if rof.dtype != numpy.float64 or not rof.flags['C_CONTIGUOUS']:
            raise ValueError("zone_set_filtered: contiguous float64 buffer required")
        self._keep.append(rof)
        self._chk(lib().fs_zone_set_filtered(self._h, ctypes.c_int(int(zone)), _pd(rof), ctypes.c_longlong(rof.size)),
                  "fs_zone_set_filtered")

    def zone_set_lu_blocks(self, zone, blocks):
        """blocks: (n, 6) ranges of interior cells (numbered from 1) that tile the zone: the blocks of the LU-SGS solve,
        as the reference's OpenMP threads take them (FastC.lu_thread_blocks); one block or None = one sweep."""
        b = numpy.ascontiguousarray(numpy.asarray([] if blocks is None else blocks, dtype=numpy.int32).reshape(-1, 6))
        self._chk(lib().fs_zone_set_lu_blocks(self._h, ctypes.c_int(int(zone)), ctypes.c_int(len(b)),
                                              b.ctypes.data_as(ctypes.c_void_p) if len(b) else None),
                  "fs_zone_set_lu_blocks")

    def set_lu_threads(self, nthreads):
        """LU-SGS / LU-SSOR solve of every zone in the blocks the reference gives that many OpenMP threads."""
        self._chk(lib().fs_set_lu_threads(self._h, ctypes.c_int(int(nthreads))), "fs_set_lu_threads")

    def zone_lu_blocks(self, zone):
        """(n, 6) LU ranges of the blocks of a zone as the library holds them (empty: one sweep)."""
        n = self._chk(lib().fs_zone_lu_blocks(self._h, ctypes.c_int(int(zone)), None, ctypes.c_int(0)), "fs_zone_lu_blocks")
        out = numpy.zeros((n, 6), dtype=numpy.int32)
        if n:
            self._chk(lib().fs_zone_lu_blocks(self._h, ctypes.c_int(int(zone)), out.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_int(n)), "fs_zone_lu_blocks")
        return out

    def set_first_it(self, first_it):
        self._chk(lib().fs_set_first_it(self._h, ctypes.c_int(int(first_it))), "fs_set_first_it")

    def zone_add(self, param_int, param_real, ro, ro_m1, ro_p1, mtr):
        for a in (ro, ro_m1, ro_p1):
            if a.dtype != numpy.float64 or not a.flags['C_CONTIGUOUS'] or a.ndim != 1:
                raise FastsHipError("zone_add: time levels must be flat contiguous float64 buffers")
        pi = numpy.ascontiguousarray(param_int, dtype=numpy.int32)
        pr = numpy.ascontiguousarray(param_real, dtype=numpy.float64)
        mt = numpy.ascontiguousarray(mtr, dtype=numpy.float64)
        self._keep += [ro, ro_m1, ro_p1]
        return self._chk(lib().fs_zone_add(self._h, _pi(pi), ctypes.c_int(pi.size), _pd(pr),
                                           ctypes.c_int(pr.size), _pd(ro), _pd(ro_m1), _pd(ro_p1),
                                           _pd(mt), ctypes.c_int(mt.size)), "fs_zone_add")

    def join_add(self, rcv_zone, donor_zone, rcv_pts, donor_pts):
        r = numpy.ascontiguousarray(rcv_pts, dtype=numpy.int32)
        d = numpy.ascontiguousarray(donor_pts, dtype=numpy.int32) if donor_pts is not None else None
        self._chk(lib().fs_join_add(self._h, ctypes.c_int(rcv_zone), ctypes.c_int(donor_zone), _pi(r),
                                    _pi(d) if d is not None else None, ctypes.c_int(r.size)), "fs_join_add")

    def send_add(self, donor_zone, peer_tag, donor_pts):
        d = numpy.ascontiguousarray(donor_pts, dtype=numpy.int32)
        self._chk(lib().fs_send_add(self._h, ctypes.c_int(donor_zone), ctypes.c_int(peer_tag), _pi(d),
                                    ctypes.c_int(d.size)), "fs_send_add")

    def zone_set_global_id(self, zone, gid):
        self._chk(lib().fs_zone_set_global_id(self._h, ctypes.c_int(zone), ctypes.c_int(gid)), "fs_zone_set_global_id")

    def send_add_remote(self, donor_zone, peer_rank, donor_pts, rcv_gid, rcv_pts):
        d = numpy.ascontiguousarray(donor_pts, dtype=numpy.int32)
        r = numpy.ascontiguousarray(rcv_pts, dtype=numpy.int32)
        if d.size != r.size:
            raise FastsHipError("send_add_remote: donor and receiver lists differ in length")
        self._chk(lib().fs_send_add_remote(self._h, ctypes.c_int(donor_zone), ctypes.c_int(peer_rank), _pi(d),
                                           ctypes.c_int(rcv_gid), _pi(r), ctypes.c_int(d.size)), "fs_send_add_remote")

    def comm_init(self, rank, world, comm_id):
        if len(comm_id) != COMM_ID_BYTES:
            raise FastsHipError("comm_init: the communicator id has %d bytes" % COMM_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._chk(lib().fs_comm_init(self._h, ctypes.c_int(rank), ctypes.c_int(world), buf, ctypes.c_int(COMM_ID_BYTES)),
                  "fs_comm_init")

    def comm_setup(self):
        self._chk(lib().fs_comm_setup(self._h), "fs_comm_setup")

    def halo_exchange(self, nstep):
        self._chk(lib().fs_halo_exchange(self._h, ctypes.c_int(nstep)), "fs_halo_exchange")

    def allreduce(self, vals, op='sum'):
        """vals reduced over the ranks of the job (after the context's stream has drained)."""
        v = numpy.ascontiguousarray(numpy.atleast_1d(vals), dtype=numpy.float64).copy()
        self._chk(lib().fs_comm_allreduce(self._h, _pd(v), ctypes.c_int(v.size), ctypes.c_int({'sum': 0, 'max': 1, 'min': 2}[op])),
                  "fs_comm_allreduce")
        return v

    def joins_set(self, pint_tc, preal_tc, rank=0):
        """Every join of the flat connectivity arrays (tc/Parameter_int, tc/Parameter_real)."""
        pi = numpy.ascontiguousarray(pint_tc, dtype=numpy.int32)
        pr = numpy.ascontiguousarray(preal_tc, dtype=numpy.float64)
        self._chk(lib().fs_joins_set(self._h, _pi(pi), ctypes.c_int(pi.size), _pd(pr), ctypes.c_int(pr.size),
                                     ctypes.c_int(int(rank))), "fs_joins_set")

    def upload(self):
        self._chk(lib().fs_upload(self._h), "fs_upload")

    def download(self, mask=FS_LEVEL_N):
        self._chk(lib().fs_download(self._h, ctypes.c_int(mask)), "fs_download")

    def compute(self, nitrun, nstep_deb, nstep_fin, nit_c=1):
        t = ctypes.c_double(0.)
        self._chk(lib().fs_compute(self._h, ctypes.c_int(nitrun), ctypes.c_int(nstep_deb),
                                   ctypes.c_int(nstep_fin), ctypes.c_int(nit_c), ctypes.byref(t)), "fs_compute")
        return t.value

    def stage(self, nitrun, nstep):
        self._chk(lib().fs_stage(self._h, ctypes.c_int(nitrun), ctypes.c_int(nstep)), "fs_stage")

    def stage_part(self, nitrun, nstep, part):
        self._chk(lib().fs_stage_part(self._h, ctypes.c_int(nitrun), ctypes.c_int(nstep), ctypes.c_int(part)),
                  "fs_stage_part")

    def set_comm_stream(self, stream_ptr):
        self._chk(lib().fs_set_comm_stream(self._h, ctypes.c_void_p(stream_ptr)), "fs_set_comm_stream")

    def halo_count(self, peer_tag, is_send):
        return self._chk(lib().fs_halo_count(self._h, ctypes.c_int(peer_tag), ctypes.c_int(int(is_send))),
                         "fs_halo_count")

    def halo_pack(self, nstep, peer_tag, dev_ptr, n):
        self._chk(lib().fs_halo_pack(self._h, ctypes.c_int(nstep), ctypes.c_int(peer_tag),
                                     ctypes.c_void_p(dev_ptr), ctypes.c_int(n)), "fs_halo_pack")

    def halo_unpack(self, nstep, peer_tag, dev_ptr, n):
        self._chk(lib().fs_halo_unpack(self._h, ctypes.c_int(nstep), ctypes.c_int(peer_tag),
                                       ctypes.c_void_p(dev_ptr), ctypes.c_int(n)), "fs_halo_unpack")

    def local_joins(self, nstep):
        self._chk(lib().fs_local_joins(self._h, ctypes.c_int(nstep)), "fs_local_joins")

    def bcs(self, nstep):
        self._chk(lib().fs_bcs(self._h, ctypes.c_int(nstep)), "fs_bcs")

    def rotate(self):
        self._chk(lib().fs_rotate(self._h), "fs_rotate")

    def fill_ghostcells(self, level=FS_LEVEL_N):
        self._chk(lib().fs_fill_ghostcells(self._h, ctypes.c_int(level)), "fs_fill_ghostcells")

    def apply_bc(self, nstep, level):
        self._chk(lib().fs_apply_bc(self._h, ctypes.c_int(nstep), ctypes.c_int(level)), "fs_apply_bc")

    def residuals(self, zone=0, neq=5):
        l2 = numpy.zeros(neq); loo = numpy.zeros(neq)
        self._chk(lib().fs_residuals(self._h, ctypes.c_int(zone), _pd(l2), _pd(loo), ctypes.c_int(neq)),
                  "fs_residuals")
        return l2, loo

    def residual_history(self, zone=0, nslot=1, neq=5):
        """(L2, Loo)[sub-iteration][equation] of the last verification step."""
        l2 = numpy.zeros((nslot, neq)); loo = numpy.zeros((nslot, neq))
        n = self._chk(lib().fs_residual_history(self._h, ctypes.c_int(zone), _pd(l2), _pd(loo), ctypes.c_int(nslot),
                                                ctypes.c_int(neq)), "fs_residual_history")
        return l2[:n], loo[:n]

    def cfl(self, zone=0):
        """[max, min, mean] local CFL number of the last verification step."""
        out = numpy.zeros(3)
        self._chk(lib().fs_cfl(self._h, ctypes.c_int(zone), _pd(out)), "fs_cfl")
        return out

    def tgv_sums(self, level=FS_LEVEL_N):
        """[enstrophy sum, kinetic-energy sum, volume] over the zones of the context."""
        out = numpy.zeros(3)
        self._chk(lib().fs_tgv_sums(self._h, ctypes.c_int(level), _pd(out)), "fs_tgv_sums")
        return out

    def tgv_sums_static(self, dev_ptr, level=FS_LEVEL_N):
        """Allocation-free form: the three sums go to the device buffer at dev_ptr (3 doubles), asynchronously."""
        self._chk(lib().fs_tgv_sums_static(self._h, ctypes.c_int(level), ctypes.c_void_p(dev_ptr)), "fs_tgv_sums_static")

    def sync(self):
        self._chk(lib().fs_sync(self._h), "fs_sync")

    def stream_triad(self, n_doubles=1 << 27, reps=20):
        """GB/s of a = b + s*c over three arrays of n_doubles doubles on this context's device."""
        g = ctypes.c_double(0.)
        self._chk(lib().fs_stream_triad(self._h, ctypes.c_longlong(int(n_doubles)), ctypes.c_int(int(reps)), ctypes.byref(g)),
                  "fs_stream_triad")
        return g.value

    def stage_kernel_ms(self, reset=False):
        avg = ctypes.c_double(0.); n = ctypes.c_int(0)
        self._chk(lib().fs_stage_kernel_ms(self._h, ctypes.c_int(int(reset)), ctypes.byref(avg),
                                           ctypes.byref(n)), "fs_stage_kernel_ms")
        return avg.value, n.value

    def debug_fetch(self, zone, which, n):
        out = numpy.zeros(n)
        self._chk(lib().fs_debug_fetch(self._h, ctypes.c_int(zone), ctypes.c_int(which), _pd(out),
                                       ctypes.c_int(n)), "fs_debug_fetch")
        return out
