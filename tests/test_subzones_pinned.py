"""Host logic of the 'implicit_local' sub-zones (fs_subzone_* of the library, no device involved) against the routines of
the compiled reference it restates, called on the same inputs: skip_lu_, ssdom_lu_ijk_ and cprdu3s1_ of
oracle/_ref/libfast_ref.so (Fast/FastC/FastC/Compute/skip_lu.for, ssdom_lu_ijk.for;
Fast/FastS/FastS/Compute/cprdu3s1.for)."""
import ctypes

import numpy as np
import pytest

from fast_amd import _capi
from oracle import ref_harness as R

pytestmark = pytest.mark.skipif(not R.available(), reason="compiled reference missing (python oracle/build_ref.py)")

_i, _d, _p = R._i, R._d, R._p


def _ptr(a, t=ctypes.c_int32):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _flush6():
    """What the Fortran runtime still holds of unit 6 goes out (the captured standard output is read after it)."""
    L = R.lib()
    L._FortranAioBeginFlush.restype = ctypes.c_void_p
    L._FortranAioBeginFlush.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    L._FortranAioEndIoStatement.argtypes = [ctypes.c_void_p]
    L._FortranAioEndIoStatement(L._FortranAioBeginFlush(6, None, 0))


@pytest.mark.parametrize("nssiter", list(range(2, 46)))
def test_skip_schedule(nssiter):
    """Both columns as init_ssiter_bloc.for:50-53 fills them: the schedule of nssiter, then that of nssiter - 1."""
    L = R.lib()
    ref = np.full(2 * nssiter, -7, dtype=np.int32)
    L.skip_lu_(_i(nssiter), _i(nssiter), _p(ref))
    L.skip_lu_(_i(nssiter - 1), _i(nssiter), _p(ref))
    got = np.full(2 * nssiter, -7, dtype=np.int32)
    lib = _capi.lib()
    assert lib.fs_subzone_skip_lu(nssiter, nssiter, _ptr(got)) == 0
    assert lib.fs_subzone_skip_lu(nssiter - 1, nssiter, _ptr(got)) == 0
    assert np.array_equal(got, ref)


def _ref_boxes(ijkv, ss, iseuil, target, nssiter=4, nitcfg=2):
    L = R.lib()
    nblk = np.array([max(1, ijkv[d] // ss[d]) for d in range(3)], dtype=np.int32)
    n = int(nblk.prod())
    no_lu = np.zeros(n + 8, dtype=np.int32)
    nis = np.zeros(nssiter, dtype=np.int32)
    ind_dm = np.zeros(6 * n * nssiter, dtype=np.int32)
    ijk_lu = np.zeros(3 * n, dtype=np.int32)
    ndfin = ctypes.c_int32(0)
    L.ssdom_lu_ijk_(_i(0), ctypes.byref(ndfin), _i(nssiter), _i(n), _i(nitcfg), _i(iseuil),
                    _p(np.asarray(ss, dtype=np.int32)), _p(np.asarray(ijkv, dtype=np.int32)), _p(nblk), _p(target),
                    _p(no_lu), _p(nis), _p(ind_dm), _p(ijk_lu))
    q = (nitcfg - 1) * 6 * n
    return ndfin.value, ind_dm[q:q + 6 * ndfin.value].copy(), no_lu[:n].copy(), int(nis[nitcfg - 1])


def _our_boxes(ijkv, ss, iseuil, target):
    lib = _capi.lib()
    n = int(np.prod([max(1, ijkv[d] // ss[d]) for d in range(3)]))
    no_lu = np.zeros(n + 8, dtype=np.int32)
    cnt = ctypes.c_int32(0)
    ind_dm = np.zeros(6 * n, dtype=np.int32)
    nb = lib.fs_subzone_boxes(_ptr(np.asarray(ijkv, dtype=np.int32)), _ptr(np.asarray(ss, dtype=np.int32)), iseuil,
                              _ptr(target), _ptr(no_lu), ctypes.byref(cnt), _ptr(ind_dm), n)
    assert nb >= 0
    return nb, ind_dm[:6 * nb].copy(), no_lu[:n].copy(), cnt.value


@pytest.mark.parametrize("ijkv,ss", [((16, 12, 12), (4, 4, 4)), ((40, 9, 7), (5, 3, 7)), ((30, 30, 1), (5, 5, 1)),
                                     ((13, 11, 17), (3, 2, 4)), ((8, 8, 8), (10, 10, 10)), ((64, 5, 5), (4, 8, 8))])
def test_boxes_of_random_targets(ijkv, ss):
    """Random targets at several densities: the boxes, their order, the block list and its length are those of
    ssdom_lu_ijk_ (isolated blocks, runs, blobs that grow into each other and boxes that merge; the cases in which the
    merge of the reference loses a box included)."""
    rng = np.random.default_rng(12)
    n = int(np.prod([max(1, ijkv[d] // ss[d]) for d in range(3)]))
    for trial in range(150):
        dens = (0.02, 0.1, 0.3, 0.6, 0.95)[trial % 5]
        target = np.where(rng.random(n) < dens, 6, 3).astype(np.int32)
        if trial % 7 == 0:      # a few solid blobs
            t3 = target.reshape([max(1, ijkv[d] // ss[d]) for d in (2, 1, 0)])
            t3[...] = 3
            for _ in range(3):
                c = [rng.integers(0, s) for s in t3.shape]
                t3[c[0]:c[0] + 2, c[1]:c[1] + 3, c[2]:c[2] + 2] = 6
            target = t3.reshape(-1)
        iseuil = 5
        rb = _ref_boxes(ijkv, ss, iseuil, target)
        ob = _our_boxes(ijkv, ss, iseuil, target)
        assert ob[0] == rb[0], (trial, ob, rb)
        assert np.array_equal(ob[1], rb[1]), (trial, ob[1], rb[1])
        assert ob[3] == rb[3]
        assert np.array_equal(ob[2][:ob[3]], rb[2][:rb[3]])


def test_boxes_when_the_reference_loses_one(capfd):
    """Cases in which the merge of the reference drops a box (it prints 'WARNING MERGE'): the same boxes here."""
    rng = np.random.default_rng(5)
    ijkv, ss = (24, 24, 24), (4, 4, 4)
    n = 216
    seen = 0
    for trial in range(400):
        target = np.where(rng.random(n) < 0.08, 6, 3).astype(np.int32)
        rb = _ref_boxes(ijkv, ss, 5, target)
        _flush6()
        out = capfd.readouterr().out
        if 'WARNING MERGE' not in out:
            continue
        seen += 1
        ob = _our_boxes(ijkv, ss, 5, target)
        assert ob[0] == rb[0] and np.array_equal(ob[1], rb[1]) and ob[3] == rb[3]
        assert np.array_equal(ob[2][:ob[3]], rb[2][:rb[3]])
    assert seen > 10


@pytest.mark.parametrize("neq,threed", [(5, 1), (6, 1), (5, 0)])
def test_block_bookkeeping(neq, threed):
    """A verification step of nssiter sub-iterations on random residual fields: counters, targets and the table of
    norms follow cprdu3s1_ (the norms of a block are reduced here in numpy: largest magnitudes exactly, sums of squares
    to rounding)."""
    L = R.lib()
    lib = _capi.lib()
    rng = np.random.default_rng(3)
    ijkv = (12, 8, 8 if threed else 1); ss = (4, 4, 4 if threed else 1)
    ific, ifick = 2, (2 if threed else 0)
    nijk = np.array([ijkv[0] + 4, ijkv[1] + 4, ijkv[2] + 2 * ifick, ific, ifick], dtype=np.int32)
    ndimdx = int(nijk[0] * nijk[1] * nijk[2])
    nblk = np.array([max(1, ijkv[d] // ss[d]) for d in range(3)], dtype=np.int32)
    n = int(nblk.prod())
    nssiter = 6
    state = {k: np.zeros(n, dtype=np.int32) for k in ('it_lu', 'it_temp', 'it_target_old')}
    state['it_target'] = rng.integers(3, 7, n).astype(np.int32)
    ours = {k: v.copy() for k, v in state.items()}
    rdm_ref = np.zeros(nssiter * neq * 2 * n); rdm_our = rdm_ref.copy()
    no_lu = np.zeros(n, dtype=np.int32)
    nis = np.zeros(nssiter, dtype=np.int32)
    it_bloc = 0
    for nstep in range(1, nssiter + 1):
        # the list of this sub-iteration: every block at the first and the last, a random subset between
        pick = np.arange(1, n + 1) if nstep in (1, nssiter) else rng.permutation(np.arange(1, n + 1))[:rng.integers(1, n)]
        no_lu[:] = 0; no_lu[:pick.size] = pick
        nis[nstep - 1] = pick.size
        it_bloc += 1
        drodm = rng.standard_normal(neq * ndimdx) * 10. ** (-nstep * rng.random())
        L.cprdu3s1_(_i(0), _i(nstep), _i(nssiter), _i(neq), _i(ndimdx), _i(n), _i(1), _i(3 if neq == 6 else 2), _i(0), _i(1),
                    _i(1), _i(0), _i(n), _i(it_bloc), _d(0.1), _p(np.asarray(ss, dtype=np.int32)), _p(nis), _p(nijk),
                    _p(np.asarray(ijkv, dtype=np.int32)), _p(nblk), _p(np.zeros(6, dtype=np.int32)),
                    _p(state['it_lu']), _p(state['it_target']), _p(state['it_target_old']), _p(state['it_temp']),
                    _p(no_lu), _p(drodm), _p(rdm_ref))
        d5 = drodm.reshape(neq, nijk[2], nijk[1], nijk[0])[:, ifick:ifick + ijkv[2], 2:2 + ijkv[1], 2:2 + ijkv[0]]
        norms = np.zeros((n, 12))
        for b in range(n):
            bi, bj, bk = b % nblk[0], (b // nblk[0]) % nblk[1], b // (nblk[0] * nblk[1])
            hi = [ijkv[d] if (bi, bj, bk)[d] == nblk[d] - 1 else ((bi, bj, bk)[d] + 1) * ss[d] for d in range(3)]
            blk = d5[:, bk * ss[2]:hi[2], bj * ss[1]:hi[1], bi * ss[0]:hi[0]].reshape(neq, -1)
            norms[b, :neq] = (blk * blk).sum(axis=1)
            norms[b, 6:6 + neq] = np.abs(blk).max(axis=1)
        rc = lib.fs_subzone_residual(_ptr(np.asarray(ijkv, dtype=np.int32)), _ptr(np.asarray(ss, dtype=np.int32)), nssiter,
                                     neq, threed, nstep, it_bloc, ctypes.c_double(0.1), _ptr(norms, ctypes.c_double),
                                     _ptr(no_lu), int(nis[nstep - 1]), _ptr(ours['it_lu']), _ptr(ours['it_target']),
                                     _ptr(ours['it_target_old']), _ptr(ours['it_temp']), _ptr(rdm_our, ctypes.c_double))
        assert rc == 0
        for k in state:
            assert np.array_equal(state[k], ours[k]), (nstep, k)
        r4 = rdm_ref.reshape(n, 2, neq, nssiter); o4 = rdm_our.reshape(n, 2, neq, nssiter)
        assert np.array_equal(r4[:, 1], o4[:, 1])                 # largest magnitudes
        assert np.allclose(r4[:, 0], o4[:, 0], rtol=1e-13, atol=0.)
        rdm_our[:] = rdm_ref      # the same table from here on: later steps compare exactly where they copy
