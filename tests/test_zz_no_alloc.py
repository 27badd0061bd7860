"""The allocation-free entry point fs_tgv_sums_static against fs_tgv_sums (same kernels, the partial sums through the
library's file-scope device array instead of the context's scratch): equal bits, on two calls in a row, the second on
a smaller zone."""
import ctypes
import numpy as np
import pytest

from fast_amd import cases, FastC, _capi
import fast_amd.PyTree as FastS
from common import SINGLE

pytestmark = pytest.mark.gpu


def _hip():
    _capi.lib()
    h = ctypes.CDLL('libamdhip64.so')
    h.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    h.hipFree.argtypes = [ctypes.c_void_p]
    return h


def test_static_scratch_variant_equals_allocating_one():
    hip = _hip()
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), 24) == 0
    try:
        for n in ((40, 22, 18), (12, 9, 7)):          # the second call reuses the array with fewer rows
            t = cases.tgv(n, 'senseur', extra_numz=SINGLE)
            (t, tc, metrics) = FastS.warmup(t, None)
            ctx = FastC.HOOK['ctx']
            for it in range(1, 3):
                FastS._compute(t, metrics, it, None)
            want = ctx.tgv_sums()
            ctx.tgv_sums_static(dev.value)
            ctx.sync()
            got = np.zeros(3)
            assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), dev, 24, 2) == 0      # device to host
            assert np.array_equal(got, want), (got, want)
            assert want[2] > 0 and want[1] > 0
    finally:
        hip.hipFree(dev)
