#!/usr/bin/env python3
"""Host<->device transfer cost of the drop-in boundary (the pyTree lives in host memory):
time of fs_upload (three time levels) and fs_download (current level) for the benchmark block,
and the throughput of the default mode in which every _compute refreshes the host copy of the
current time level.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fast_amd import cases, FastC, Internal
import fast_amd.PyTree as FastS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = cases.tgv(n, 'senseur')
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
z = Internal.getZones(t)[0]
nbytes = FastC.level_buffer(z).nbytes
ctx.sync()
t0 = time.perf_counter(); ctx.upload(); ctx.sync(); up = time.perf_counter() - t0
t0 = time.perf_counter(); ctx.download(1); ctx.sync(); down = time.perf_counter() - t0
for it in range(1, 3):
    FastS._compute(t, metrics, it, None)
t0 = time.perf_counter()
K = 5
for it in range(3, 3 + K):
    FastS._compute(t, metrics, it, None)          # step + download of the current level
ctx.sync()
el = (time.perf_counter() - t0) / K
print(json.dumps({"cells": n ** 3, "level_bytes": nbytes,
                  "upload_3_levels_s": up, "upload_GBs": 3 * nbytes / up / 1e9,
                  "download_1_level_s": down, "download_GBs": nbytes / down / 1e9,
                  "step_plus_download_ms": el * 1e3,
                  "Mcell_iter_s_with_host_refresh": n ** 3 / el / 1e6}))
