"""Implicit multi-block job on one GPU: LU-SGS sweeps of the zones on one stream or on several.
2 x 2 x 2 blocks of 64^3 cells, ausmpred, 3 Newton sub-iterations; run with FASTS_HIP_LU_STREAMS=1 and =4."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fast_amd import cases, connect, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS

blocks = (2, 2, 2); n = 64
nn = tuple(n * b for b in blocks)
t = cases.tgv(nn, 'ausmpred', blocks=blocks, temporal_scheme='implicit', time_step=0.005,
              extra_numb={'ss_iteration': 3}, modulo_verif=1000)
zones = Internal.getZones(t)
tc = C.node2Center(t)
connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, (n, n, n)), [z[0] for z in zones])
(t, tc, metrics) = FastS.warmup(t, tc)
os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
for it in range(1, 4):
    FastS._compute(t, metrics, it, tc)
torch.cuda.synchronize()
t0 = time.perf_counter()
nst = 10
for it in range(4, 4 + nst):
    FastS._compute(t, metrics, it, tc)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nst
print(json.dumps({"lu_streams": os.environ.get('FASTS_HIP_LU_STREAMS', '4'), "blocks": blocks, "cells_per_block": n ** 3,
                  "ms_per_step": dt * 1e3, "ms_per_sub_iteration": dt * 1e3 / 3}))
