#!/usr/bin/env python3
"""Golden vectors of the 'implicit_local' scheme (tests/golden/local_*.npz) from the COMPILED REFERENCE
(oracle/_ref/libfast_ref.so driven by oracle/ref_harness.py: init_ssiter_bloc_, navier_stokes_struct_, cprdu3s1_, invlu_
and the bvbs_* routines called once per sub-zone box, as the reference's task loops call them).

    python tests/golden/make_golden_local.py [case names]

Each file holds the inputs (Parameter_int, Parameter_real, metrics, initial field, join tables) and, for every time step:
the boxes of every sub-iteration and zone (`boxes_it<n>`: [nssiter, nzone, max, 6], -1 padded, with their counts in
`nbox_it<n>`), the targets of the blocks after the step (`target_it<n>_<zone>`), the field of every zone
(`ro_it<n>_<zone>`) and, on verification steps, the norms of the sub-iterations that ran (`hist_it<n>_<zone>`:
[slots, 2, neq], L2 then Loo, assembled from the table of cprdu3s1_ like display_ss_iteration.cpp does).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
from fast_amd import cases, connect, FastC, Internal
from oracle import ref_harness as R
from oracle.oracle import assemble_rdm
from common import prepare, zone_arrays, wall_distance

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}

# name: dict(kind, n, scheme, model, blocks, bcs, perturb, ss_iteration, kmod (modulo_verif), ssdom, epsi, nit, extra)
CASES = {
    'local_tgv_ausmpred':     dict(kind='tgv', n=(16, 12, 12), scheme='ausmpred', model='NSLaminar', ss_iteration=6, kmod=1,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=8),
    'local_tgv_senseur_k2':   dict(kind='tgv', n=(16, 12, 12), scheme='senseur', model='NSLaminar', ss_iteration=6, kmod=2,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=9),
    'local_tgv_euler_k3':     dict(kind='tgv', n=(20, 8, 12), scheme='roe', model='Euler', ss_iteration=8, kmod=3,
                                   ssdom=(5, 4, 3), epsi=0.05, nit=10),
    'local_channel_curv':     dict(kind='tgv', n=(16, 12, 12), scheme='ausmpred', model='NSLaminar', ss_iteration=6, kmod=2,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=8, perturb=0.15,
                                   bcs={'jmin': 'BCWall', 'jmax': 'BCWallViscous', 'kmin': 'BCFarfield'}),
    'local_walls_cart':       dict(kind='tgv', n=(16, 12, 12), scheme='senseur', model='NSLaminar', ss_iteration=7, kmod=2,
                                   ssdom=(4, 3, 4), epsi=0.1, nit=8,
                                   bcs={'imin': 'BCWall', 'jmax': 'BCWallInviscid', 'kmax': 'BCExtrapolate'}),
    'local_sa_curv':          dict(kind='tgv', n=(12, 12, 8), scheme='ausmpred', model='NSTurbulent', ss_iteration=6, kmod=2,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=8, perturb=0.15, bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
    'local_sa_cart':          dict(kind='tgv', n=(12, 12, 8), scheme='ausmpred', model='NSTurbulent', ss_iteration=6, kmod=2,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=7, bcs={'jmin': 'BCWall', 'jmax': 'BCWallViscous'}),
    'local_two_blocks':       dict(kind='tgv', n=(12, 12, 24), scheme='ausmpred', model='NSLaminar', ss_iteration=6, kmod=1,
                                   ssdom=(4, 4, 4), epsi=0.1, nit=6, blocks=(1, 1, 2)),
    # wider than one tile of the Cartesian stage kernel (64 x 8 cells) and than one patch of the curvilinear one (31 x 7):
    # the boxes launch tile lists; the field is kept for the last step only
    'local_tiles_cart':       dict(kind='tgv', n=(72, 20, 16), scheme='senseur', model='NSLaminar', ss_iteration=6, kmod=2,
                                   ssdom=(12, 5, 4), epsi=0.1, nit=6, last_only=True),
    'local_tiles_curv':       dict(kind='tgv', n=(40, 18, 20), scheme='ausmpred', model='NSLaminar', ss_iteration=6, kmod=2,
                                   ssdom=(8, 6, 5), epsi=0.1, nit=6, perturb=0.15, last_only=True,
                                   bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
    'local_plane2d':          dict(kind='tgv2d', n=(24, 16), scheme='ausmpred', model='NSLaminar', ss_iteration=6, kmod=2,
                                   ssdom=(4, 4, 1), epsi=0.1, nit=10, perturb=0.15, bcs={'jmin': 'BCWall', 'jmax': 'BCFarfield'}),
}


def build(c):
    """The tree of one case; shared with the tests that replay the golden vectors."""
    extra = dict(SINGLE, ssdom_IJK=list(c['ssdom']), epsi_newton=c['epsi'])
    numb = {'ss_iteration': c['ss_iteration']}
    if c['kind'] == 'tgv':
        return cases.tgv(c['n'], c['scheme'], model=c['model'], blocks=c.get('blocks', (1, 1, 1)), extra_numz=extra,
                         bcs=c.get('bcs'), modulo_verif=c['kmod'], perturb=c.get('perturb', 0.), time_step=0.02,
                         temporal_scheme='implicit_local', extra_numb=numb)
    numb['modulo_verif'] = c['kmod']
    return cases.tgv2d(c['n'], c['scheme'], model=c['model'], extra_numz=extra, bcs=c.get('bcs'),
                       perturb=c.get('perturb', 0.), time_step=0.02, temporal_scheme='implicit_local', extra_numb=numb)


def case_joins(c):
    blocks = c.get('blocks', (1, 1, 1))
    if blocks == (1, 1, 1):
        return []
    nv = tuple(c['n'][d] // blocks[d] for d in range(3))
    return connect.periodic_block_joins(blocks, nv)


def main():
    if not R.available():
        raise SystemExit("make_golden_local: oracle/_ref/libfast_ref.so is missing (python oracle/build_ref.py)")
    only = sys.argv[1:]
    for name, c in CASES.items():
        if only and name not in only:
            continue
        t = build(c)
        metrics = prepare(t)
        zones = Internal.getZones(t)
        nb = len(zones)
        nit = c['nit']
        out = {'nzone': np.int32(nb), 'nit': np.int32(nit)}
        rz = []
        for b, (z, m) in enumerate(zip(zones, metrics)):
            pint, preal, xyz = zone_arrays(z)
            ro0 = FastC.level_buffer(z).copy()
            dist = wall_distance(z)
            rz.append(R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], ro0, m[0], dist))
            out['pint%d' % b] = np.asarray(pint, dtype=np.int32)
            out['preal%d' % b] = np.asarray(preal, dtype=np.float64)
            out['mtr%d' % b] = m[0]
        joins = [(br, r, bd, d) for br, bd, r, d in case_joins(c)]
        s = R.RefSolver(rz, joins)
        s.local = True
        s.kmod = c['kmod']
        s.fill_ghostcells('ro')
        for b in range(nb):
            out['ghost_%d' % b] = rz[b].ro.copy()
            rz[b].ro_m1[:] = rz[b].ro; rz[b].ro_p1[:] = rz[b].ro
        nssiter = int(rz[0].pint[58])
        shapes = []
        for it in range(1, nit + 1):
            s.step(it)
            mx = max([1] + [len(bz) for bs in s.last_boxes for bz in bs])
            boxes = np.full((nssiter, nb, mx, 6), -1, dtype=np.int32)
            nbox = np.zeros((nssiter, nb), dtype=np.int32)
            for q, bs in enumerate(s.last_boxes):
                for b, bz in enumerate(bs):
                    nbox[q, b] = len(bz)
                    for r, x in enumerate(bz):
                        boxes[q, b, r] = x
            out['boxes_it%d' % it] = boxes
            out['nbox_it%d' % it] = nbox
            shapes.append([int(v) for v in nbox.sum(axis=1)])
            for b, zn in enumerate(rz):
                out['target_it%d_%d' % (it, b)] = s._sz_views(zn)['it_target'][:zn.sz_n].copy()
                if not c.get('last_only') or it == nit:
                    out['ro_it%d_%d' % (it, b)] = zn.ro.copy()
                if it % c['kmod'] == 0:
                    l2, loo = assemble_rdm(zn.pint, zn.preal, zn.rdm, nssiter, zn.it_bloc)
                    out['hist_it%d_%d' % (it, b)] = np.stack([l2, loo], axis=1)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print("%-24s %7.1f KB   boxes per sub-iteration and step: %s" % (name, os.path.getsize(path) / 1024., shapes))


if __name__ == '__main__':
    main()
