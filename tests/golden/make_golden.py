#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/*.npz from the COMPILED REFERENCE
(oracle/_ref/libfast_ref.so, built by oracle/build_ref.py from the Fortran sources where they
lie in the reference tree) driven by oracle/ref_harness.py.

    python tests/golden/make_golden.py [case names]

Each file holds the inputs (Parameter_int, Parameter_real, metrics, initial field without
ghost fill, join tables) and the reference's outputs (field after 1, 3 and `nit` time steps and
the per-step residual norms of the first sub-iteration; for the implicit scheme `reshist` holds the norms of
every Newton sub-iteration of every step, out of the reference's own cprdu3s1 with every step treated as a
verification step, `disp_text` what the reference's dpssiter_ prints after each step, `disp_text_long_all` the long
format with every sub-iteration after the last step, and `zch_*` the ZoneConvergenceHistory arrays that its
conv2pytree_ has filled by then).  Cases marked 'single' run the
reference with one cache block per zone (cache_blocking_I/J/K larger than the zone), the
configuration in which its floating-point operation order is reproducible by a plain loop nest;
'default' keeps the reference's default cache blocking (2048,2,7), whose results differ from
'single' in the last bits (or, for the sensor scheme, by sensor flips).
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
from common import prepare, zone_arrays, wall_distance, data_bcs

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}

# name: (kind, n, scheme, model, blocks, cache, nit)
CASES = {
    'tgv12_senseur_lamin':      ('tgv', 12, 'senseur', 'NSLaminar', (1, 1, 1), 'single', 100),
    'tgv12_ausmpred_lamin':     ('tgv', 12, 'ausmpred', 'NSLaminar', (1, 1, 1), 'single', 100),
    'tgv12_roe_lamin':          ('tgv', 12, 'roe', 'NSLaminar', (1, 1, 1), 'single', 10),
    'tgv10_senseur_euler':      ('tgv', 10, 'senseur', 'Euler', (1, 1, 1), 'single', 10),
    'vortex12_roe_euler':       ('vortex', 12, 'roe', None, (1, 1, 1), 'single', 100),
    'tgv_ragged_senseur_lamin': ('tgv', (13, 6, 9), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 5),
    'tgv_2x2_senseur_lamin':    ('tgv', (8, 12, 20), 'senseur', 'NSLaminar', (1, 2, 2), 'single', 10),
    'channel_senseur_lamin':    ('tgv', (12, 10, 8), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 20,
                                 {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}),
    'box_farfield_ausmpred':    ('tgv', (9, 8, 10), 'ausmpred', 'NSLaminar', (1, 1, 1), 'single', 20,
                                 {f: 'BCFarfield' for f in ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')}),
    'box_mixed_bc_euler':       ('tgv', (8, 9, 7), 'ausmpred', 'Euler', (1, 1, 1), 'single', 10,
                                 {'imin': 'BCExtrapolate', 'imax': 'BCFarfield', 'jmin': 'BCWall',
                                  'jmax': 'BCSymmetryPlane', 'kmin': 'BCWallInviscid'}),
    # curvilinear zones (perturbed mesh) and implicit LU-SGS steps: a dict of options after the BCs
    'curv_senseur_lamin':       ('tgv', (12, 10, 9), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 20, None,
                                 {'perturb': 0.15}),
    'curv_channel_ausmpred':    ('tgv', (10, 9, 8), 'ausmpred', 'NSLaminar', (1, 2, 1), 'single', 10,
                                 {'imin': 'BCFarfield', 'imax': 'BCFarfield', 'kmin': 'BCWallInviscid', 'kmax': 'BCWall'},
                                 {'perturb': 0.15}),
    'implicit_ausmpred_lamin':  ('tgv', 12, 'ausmpred', 'NSLaminar', (1, 1, 1), 'single', 10, None,
                                 {'ss_iteration': 5, 'time_step': 0.02}),
    'implicit_curv_senseur':    ('tgv', (10, 9, 8), 'senseur', 'NSLaminar', (1, 1, 2), 'single', 6, None,
                                 {'ss_iteration': 3, 'time_step': 0.02, 'perturb': 0.15}),
    'implicit_channel_curv':    ('tgv', (10, 9, 8), 'ausmpred', 'NSLaminar', (1, 1, 1), 'single', 8,
                                 {'jmin': 'BCWall', 'jmax': 'BCWallViscous', 'kmin': 'BCFarfield'},
                                 {'ss_iteration': 4, 'time_step': 0.02, 'perturb': 0.15}),
    # Spalart-Allmaras (6 equations; the wall distance is stored with the inputs), two-dimensional zones
    # (kind 'tgv2d', n = (ni, nj)) and the NACA0012 C-mesh (kind 'naca', n = (nfoil, nwake, nj)): steady-RANS
    # set-up of the reference, SA + implicit LU-SGS + local time step + wall / far field / wake match join
    'sa_channel_ausmpred':      ('tgv', (10, 9, 8), 'ausmpred', 'NSTurbulent', (1, 1, 1), 'single', 10,
                                 {'jmin': 'BCWall', 'jmax': 'BCWallViscous'}),
    'sa_implicit_curv_roe':     ('tgv', (10, 9, 8), 'roe', 'NSTurbulent', (1, 1, 1), 'single', 6, None,
                                 {'ss_iteration': 3, 'time_step': 0.02, 'perturb': 0.15}),
    'plane2d_senseur_lamin':    ('tgv2d', (14, 10), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 20,
                                 {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {'perturb': 0.15}),
    'naca0012_sa_implicit':     ('naca', (16, 6, 10), 'ausmpred', 'NSTurbulent', (1, 1, 1), 'single', 20),
    # BCs that read window data from Parameter_real: two pressure outlets, one isothermal wall, one far field
    'outlet_isowall_curv':      ('tgv', (10, 9, 8), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 10,
                                 data_bcs((10, 9, 8), inj=False), {'perturb': 0.15}),
    # explicit scheme with a local time step (roe + walls); walls with flat ghost cells behind them and a RANS/LES
    # extrapolation of nu_tilde, implicit
    'explicit_local_roe_curv':  ('tgv', (10, 9, 8), 'roe', 'NSLaminar', (1, 1, 1), 'single', 10,
                                 {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {'perturb': 0.15, 'local': 0.5}),
    'collapsed_walls_sa_impl':  ('tgv', (10, 9, 8), 'ausmpred', 'NSTurbulent', (1, 1, 1), 'single', 6,
                                 {'jmin': 'BCWall', 'kmax': 'BCWall', 'imin': 'BCExtrapolateRANS'},
                                 {'perturb': 0.15, 'collapse': ('jmin', 'kmax'), 'ss_iteration': 3, 'time_step': 0.02}),
    # slope 'o3sc': the o3 states weighted by the shock sensor of SENSOR/sciacovelli.for; 'front' puts a compression
    # front across the zone so that the sensor departs from 1
    'o3sc_senseur_walls':       ('tgv', (14, 10, 9), 'senseur', 'NSLaminar', (1, 1, 1), 'single', 10,
                                 {'jmin': 'BCWall', 'jmax': 'BCFarfield'}, {'slope': 'o3sc', 'front': 1}),
    'o3sc_ausmpred_curv':       ('tgv', (12, 10, 9), 'ausmpred', 'NSLaminar', (1, 1, 1), 'single', 10,
                                 {'kmin': 'BCWall'}, {'slope': 'o3sc', 'front': 1, 'perturb': 0.15}),
    'o3sc_roe_implicit_curv':   ('tgv', (12, 10, 9), 'roe', 'Euler', (1, 1, 1), 'single', 6, None,
                                 {'slope': 'o3sc', 'front': 1, 'perturb': 0.15, 'ss_iteration': 3, 'time_step': 0.02}),
    'tgv12_ausmpred_default':   ('tgv', 12, 'ausmpred', 'NSLaminar', (1, 1, 1), 'default', 100),
    'tgv12_senseur_default':    ('tgv', 12, 'senseur', 'NSLaminar', (1, 1, 1), 'default', 100),
}


def build(kind, n, scheme, model, blocks, cache, bcs=None, opts=None, modulo_verif=50):
    """The tree of one case; shared with the tests that replay the golden vectors."""
    extra = SINGLE if cache == 'single' else None
    opts = opts or {}
    if opts.get('local'):      # time_step_nature='local' at the given CFL number
        extra = dict(extra or {}, time_step_nature='local', cfl=opts['local'])
    if opts.get('slope'):
        extra = dict(extra or {}, slope=opts['slope'])
    if kind == 'tgv':
        kw = {}
        if 'collapse' in opts:
            kw['collapse'] = opts['collapse']
        if 'ss_iteration' in opts:
            kw.update({'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': opts['ss_iteration']}})
        t = cases.tgv(n, scheme, model=model, blocks=blocks, extra_numz=extra, bcs=bcs, modulo_verif=modulo_verif,
                      perturb=opts.get('perturb', 0.), time_step=opts.get('time_step'), **kw)
        if opts.get('front'):      # u falls by 4 over a few cells along i
            for z in Internal.getZones(t):
                u = Internal.getNodeFromName2(z, 'VelocityX')[1]
                u -= 2. * np.tanh((np.arange(u.shape[0]) - u.shape[0] / 2.) / 1.5)[:, None, None]
        return t
    if kind == 'tgv2d':
        kw = {}
        if 'ss_iteration' in opts:
            kw = {'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': opts['ss_iteration']}}
        return cases.tgv2d(n, scheme, model=model, extra_numz=extra, bcs=bcs, perturb=opts.get('perturb', 0.),
                           time_step=opts.get('time_step'), extra_numb={'modulo_verif': modulo_verif}, **kw)
    if kind == 'naca':
        return cases.naca_cmesh(nfoil=n[0], nwake=n[1], nj=n[2], scheme=scheme, model=model, extra_numz=extra,
                                extra_numb={'modulo_verif': modulo_verif})
    return cases.vortex(n, scheme, extra_numz=extra)


def case_joins(kind, n, blocks):
    """[(receiver_zone, donor_zone, receiver_pts, donor_pts)] of a case."""
    if kind == 'naca':
        rp, dp = connect.cmesh_wake_join(2 * n[0] + 2 * n[1], n[2], n[1])
        return [(0, 0, rp, dp)]
    if blocks == (1, 1, 1):
        return []
    if isinstance(n, int):
        n = (n, n, n)
    nv = tuple(n[d] // blocks[d] for d in range(3))
    return connect.periodic_block_joins(blocks, nv)


def main():
    if not R.available():
        raise SystemExit("make_golden: oracle/_ref/libfast_ref.so is missing (python oracle/build_ref.py)")
    only = sys.argv[1:]          # optional: names of the cases to (re)generate
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        (kind, n, scheme, model, blocks, cache, nit) = spec[:7]
        t = build(kind, n, scheme, model, blocks, cache, spec[7] if len(spec) > 7 else None,
                  spec[8] if len(spec) > 8 else None)
        metrics = prepare(t)
        zones = Internal.getZones(t)
        nb = len(zones)
        out = {'nzone': np.int32(nb), 'nit': np.int32(nit)}
        rz = []
        for b, (z, m) in enumerate(zip(zones, metrics)):
            pint, preal, xyz = zone_arrays(z)
            ro0 = FastC.level_buffer(z).copy()
            dist = wall_distance(z)
            rz.append(R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], ro0, m[0], dist))
            if dist is not None:
                out['dist%d' % b] = np.ascontiguousarray(dist.reshape(-1, order='F'))
            # metrics from the reference's own cp_tijk / cp_vol, then tijk_extrap on the collapsed ghost cells
            mref = R.ref_metrics(pint, preal, xyz[0], xyz[1], xyz[2])
            if int(pint[25]) in (0, 3):
                mref = R.ref_tijk_extrap(pint, R.ref_degenerate(xyz[0], xyz[1], xyz[2], int(pint[3]), int(pint[4])), mref)
            assert np.array_equal(mref, m[0])
            out['mtr%d' % b] = mref
            out['pint%d' % b] = np.asarray(pint, dtype=np.int32)
            out['preal%d' % b] = np.asarray(preal, dtype=np.float64)
            out['ro0_%d' % b] = ro0
        joins = [(br, r, bd, d) for br, bd, r, d in case_joins(kind, n, blocks)]
        if joins:
            out['njoin'] = np.int32(len(joins))
            for q, (br, r, bd, d) in enumerate(joins):
                out['join%d_zones' % q] = np.array([br, bd], dtype=np.int32)
                out['join%d_rcv' % q] = r
                out['join%d_dnr' % q] = d
        s = R.RefSolver(rz, joins)
        implicit = int(rz[0].pint[29]) == 1
        s.verif = implicit
        s.fill_ghostcells('ro')
        for b in range(nb):
            out['ghost_%d' % b] = rz[b].ro.copy()
            rz[b].ro_m1[:] = rz[b].ro; rz[b].ro_p1[:] = rz[b].ro      # the three levels start equal
        res = np.zeros((nit, nb, 2, 5))
        nssiter = int(rz[0].pint[58]); neq = rz[0].neq
        hist = np.zeros((nit, nb, nssiter, 2, neq))
        # convergence history and printed table of the reference's own display routines (dpssiter_, conv2pytree_)
        nrec = nit + 2
        zch = [dict(last=np.zeros(1, dtype=np.int32), itnum=np.zeros(nrec, dtype=np.int32), l2=np.zeros(neq * nrec),
                    oo=np.zeros(neq * nrec), l2_diff=np.zeros(neq * nrec), oo_diff=np.zeros(neq * nrec)) for _ in range(nb)]
        disp = []
        znames = [z[0] for z in zones]
        for it in range(1, nit + 1):
            s.step(it)
            if implicit:
                disp.append(s.display_reference(it, znames, zch, lft=1, iverb=0))
                if it == nit:
                    out['disp_text_long_all'] = np.array(s.display_reference(it, znames, zch, lft=0, iverb=2, record=False))
            if implicit:     # norms of every Newton sub-iteration, from cprdu3s1_ of the compiled reference
                for b, (l2, loo) in enumerate(s.residual_history()):
                    hist[it - 1, b, :, 0] = l2; hist[it - 1, b, :, 1] = loo
                    res[it - 1, b, 0] = l2[0, :5]; res[it - 1, b, 1] = loo[0, :5]
            else:            # explicit scheme: the reference tabulates nothing; the same norm of the first-stage increment
                for b, (l2, loo) in enumerate(s.residuals()):
                    res[it - 1, b, 0] = l2[:5]; res[it - 1, b, 1] = loo[:5]
            if it in (1, 3, nit):
                for b in range(nb):
                    out['ro_it%d_%d' % (it, b)] = rz[b].ro.copy()
        out['residuals'] = res
        if kind == 'tgv' and all(int(z.pint[25]) == 2 for z in rz):
            # kinetic energy / enstrophy sums of the final state, per zone, from post_ of the compiled reference
            out['tgvsums'] = np.array(s.tgv_sums())
        if implicit:
            out['reshist'] = hist
            out['disp_text'] = np.array(disp)
            out['zone_names'] = np.array(znames)
            for b in range(nb):
                for key in ('last', 'itnum', 'l2', 'oo', 'l2_diff', 'oo_diff'):
                    out['zch_%s_%d' % (key, b)] = zch[b][key]
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.))


if __name__ == '__main__':
    main()
