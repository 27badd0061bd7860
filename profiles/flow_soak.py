#!/usr/bin/env python3
"""Soak of the one-launch LU-SGS sweeps: long implicit runs on several zone shapes, the sweeps as one launch each against one
launch per hyperplane, fields compared bit for bit at the end.  python profiles/flow_soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
CASES = [
    ('tgv 256^3 senseur', dict(n=256, scheme='senseur', model='NSLaminar', perturb=0., bcs=None, blocks=(1, 1, 1), numz={})),
    ('curvilinear 160x120x96 walls', dict(n=(160, 120, 96), scheme='ausmpred', model='NSLaminar', perturb=0.15,
                                           bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'}, blocks=(1, 1, 1), numz={})),
    ('SA 96x80x64 walls', dict(n=(96, 80, 64), scheme='roe', model='NSTurbulent', perturb=0., bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'},
                               blocks=(1, 1, 1), numz={})),
    ('SSOR 2 relaxations 128x96x80', dict(n=(128, 96, 80), scheme='senseur', model='NSLaminar', perturb=0.1,
                                          bcs={'imin': 'BCWall', 'imax': 'BCFarfield'}, blocks=(1, 1, 1), numz={'nb_relax': 2})),
    ('8 blocks of 64x48x40, joins', dict(n=(128, 96, 80), scheme='senseur', model='NSLaminar', perturb=0.15,
                                         bcs={'imin': 'BCWall', 'imax': 'BCFarfield'}, blocks=(1, 2, 4), numz={})),
]

def run(c, flow):
    os.environ['FASTS_HIP_LU_FLOW'] = str(flow)
    nn = c['n'] if isinstance(c['n'], tuple) else (c['n'],) * 3
    t = cases.tgv(c['n'], c['scheme'], model=c['model'], blocks=c['blocks'], extra_numz=dict(SINGLE, **c['numz']), bcs=c['bcs'],
                  perturb=c['perturb'], time_step=0.003 * 32. / max(nn) * 10., temporal_scheme='implicit', extra_numb={'ss_iteration': 3})
    tc = None
    zones = Internal.getZones(t)
    if c['blocks'] != (1, 1, 1):
        nv = tuple(nn[d] // c['blocks'][d] for d in range(3))
        tc = C.node2Center(t)
        connect.add_joins_to_tc(tc, connect.periodic_block_joins(c['blocks'], nv), [z[0] for z in zones])
    (t, tc, metrics) = FastS.warmup(t, tc)
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        FastS._compute(t, metrics, it, tc)
    out = [FastC.level_buffer(z).copy() for z in Internal.getZones(t)]
    el = time.perf_counter() - t0
    FastC.HOOK['ctx'].close()
    return out, el

bad = 0
for name, c in CASES:
    ref, t0 = run(c, 0)
    got, t1 = run(c, 1)
    ok = all(np.array_equal(a, b) for a, b in zip(ref, got)) and all(np.isfinite(a).all() for a in ref)
    bad += not ok
    print("%-36s %3d steps: one launch per hyperplane %.1f ms / per sweep %.1f ms per step (host copy refreshed each step); fields %s"
          % (name, steps, t0 / steps * 1e3, t1 / steps * 1e3, "identical" if ok else "DIFFER"), flush=True)
print("soak: %d of %d cases differ" % (bad, len(CASES)))
sys.exit(1 if bad else 0)
