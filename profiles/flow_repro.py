#!/usr/bin/env python3
"""Development aid: the four-block curvilinear wall / far-field job of tests/test_gpu_mpi.py (walls_implicit) in one process,
repeated, against the CPU restatement; prints where the device differs.  FASTS_HIP_LU_FLOW / _WAVES as in the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from oracle import oracle as O
from common import zone_arrays, prepare

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
blocks = (1, 2, 2); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
layer = sys.argv[3] if len(sys.argv) > 3 else 'c'
nn = tuple(n * b for b in blocks)
opts = dict(perturb=0.15, bcs={'imin': 'BCWall', 'imax': 'BCFarfield'}, time_step=0.01, temporal_scheme='implicit',
            extra_numb={'ss_iteration': 3}, extra_numz=SINGLE)

def oracle():
    t = cases.tgv(nn, 'senseur', model='NSLaminar', blocks=blocks, **opts)
    m = prepare(t)
    zs = []
    for z, mm in zip(Internal.getZones(t), m):
        p, pr, _ = zone_arrays(z)
        zs.append(O.OracleZone(p, pr, FastC.level_buffer(z), mm[0]))
    s = O.OracleSolver(zs, [(br, r, bd, d) for br, bd, r, d in connect.periodic_block_joins(blocks, (n, n, n))])
    s.fill_ghostcells()
    for it in range(3):
        s.step()
    return [z.ro.copy() for z in s.zones], zs[0].pint

ref, pint = oracle()
bad = 0
for rep in range(reps):
    t = cases.tgv(nn, 'senseur', model='NSLaminar', blocks=blocks, **opts)
    zones = Internal.getZones(t)
    tc = C.node2Center(t)
    connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, (n, n, n)), [z[0] for z in zones])
    (t, tc, metrics) = FastS.warmup(t, tc)
    zones = Internal.getZones(t)
    for it in range(1, 4):
        FastS._compute(t, metrics, it, tc, layer=layer)
    wrong = []
    for b, z in enumerate(zones):
        got = FastC.level_buffer(z)
        if not np.array_equal(got, ref[b]):
            d = np.abs(got - ref[b]).reshape(5, -1)
            idx = np.argwhere(d.max(axis=0) > 0).ravel()
            wrong.append((b, float(d.max()), idx.size, idx[:6].tolist()))
    FastC.HOOK['ctx'].close()
    if wrong:
        bad += 1
        print("rep %d: %s" % (rep, wrong))
print("FLOW=%s WAVES=%s layer=%s: %d of %d runs differ" % (os.environ.get('FASTS_HIP_LU_FLOW'), os.environ.get('FASTS_HIP_LU_FLOW_WAVES'), layer, bad, reps))
