#!/usr/bin/env python3
"""Development aid: where the time of one step of a one-launch LU-SGS sweep goes, from wall-clock stamps (100 MHz) taken
by the wave that works the task in the middle of the order of the lower sweep (library built with -DFS_LU_STAMP):

    FASTS_HIP_LIB=fast_amd/csrc/libfasts_hip_lustamp.so python profiles/flow_stamps.py [cells]

Slots per step: 0 top of the step, 1 predecessors' step counts seen (own count published), 2 increments of the
neighbours landed and their terms added, 3 recurrence done, 4 stores issued.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_amd import cases, FastC, _capi
import fast_amd.PyTree as FastS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
t = cases.tgv(n, 'senseur', time_step=0.003 * 32. / n * 10., temporal_scheme='implicit', extra_numb={'ss_iteration': 3})
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
for it in range(1, 3):
    ctx.compute(it, 1, 4, 1)
ctx.sync()
buf = np.zeros(1024 * 8, dtype=np.int64)
rc = _capi.lib().fs_lu_stamp_fetch(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
nsteps = n // 8 + 7
s = buf.reshape(1024, 8)[:nsteps + 1, :5].astype(float) * 10.   # ns
names = ['wait for the predecessors', 'increments + neighbour terms', 'recurrence', 'stores issued', 'to the next step']
d = np.diff(s[:nsteps], axis=1)
nxt = s[1:nsteps + 1, 0] - s[:nsteps, 4]
print("steps %d, one step of the wave: mean %.0f ns (min %.0f, max %.0f)" % (nsteps, np.diff(s[:nsteps + 1, 0]).mean(),
      np.diff(s[:nsteps + 1, 0]).min(), np.diff(s[:nsteps + 1, 0]).max()))
for q in range(4):
    print("  %-32s mean %6.0f ns  median %6.0f  max %6.0f" % (names[q], d[:, q].mean(), np.median(d[:, q]), d[:, q].max()))
print("  %-32s mean %6.0f ns  median %6.0f  max %6.0f" % (names[4], nxt.mean(), np.median(nxt), nxt.max()))
