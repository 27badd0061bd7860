#!/usr/bin/env python3
"""Development aid: where the time of one lower-sweep launch of the LU-SGS solve goes, from stamps taken by wave
(0,0) of every launch (library built with -DFS_LU_STAMP, see fasts_lusgs.hip):

    FASTS_HIP_LIB=fast_amd/csrc/libfasts_hip_lustamp.so python profiles/lu_stamps.py [cells]

Slots: 0 wall clock at entry (100 MHz), 1 shader clock at entry, 2 after every load has landed and the j / k neighbour
terms and the matrix of the i predecessor are done, 3 after the 8-step recurrence, 4 after the stores have completed,
7 wall clock at the end.  Wave (0,0) is the first wave of the launch, so wall(entry of h+1) - wall(end of h) is the
hand-over between two dependent launches as the wave sees it.
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
s = buf.reshape(1024, 8)
nh = n // 8 + 2 * n - 2
s = s[:nh]
wall = 10.0   # ns per wall-clock tick
ent, end = s[:, 0] * wall, s[:, 7] * wall
print("hyperplanes", nh, " sweep by wall clock: %.1f us" % ((end[-1] - ent[0]) / 1e3))
inside = end - ent
handover = ent[1:] - end[:-1]
print("inside the wave (entry -> stores complete): mean %.0f ns, min %.0f, max %.0f" % (inside.mean(), inside.min(), inside.max()))
print("hand-over (end of wave (0,0) of h -> entry of wave (0,0) of h+1): mean %.0f ns, min %.0f, max %.0f" %
      (handover.mean(), handover.min(), handover.max()))
cyc = s[:, 1:5].astype(float)
ph = np.diff(cyc, axis=1)
tot = ph.sum(axis=1)
scale = inside / np.maximum(tot, 1.)   # ns per shader-clock tick, per launch
print("ns per shader tick (median): %.3f" % np.median(scale))
names = ['loads + j/k terms + i matrix', 'recurrence', 'stores']
for lo, hi in ((0, 50), (50, nh // 2 - 50), (nh // 2 - 50, nh // 2 + 50), (nh - 100, nh)):
    m = ph[lo:hi].mean(axis=0) * np.median(scale)
    print("h %4d..%4d: " % (lo, hi) + " | ".join("%s %6.0f ns" % (nm, v) for nm, v in zip(names, m)) +
          " | hand-over %6.0f ns" % handover[lo:hi - 1].mean())
