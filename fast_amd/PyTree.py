"""MI355X counterpart of ``FastS.PyTree``: same entry points, same pyTree in / out.

    (t, tc, metrics) = warmup(t, tc)              Fast/FastS/FastS/PyTree.py:332-536
    _compute(t, metrics, nitrun, tc)              Fast/FastS/FastS/PyTree.py:42-241
    _applyBC / _fillGhostcells                    Fast/FastS/FastS/PyTree.py:1146-1218
    display_temporal_criteria                     Fast/FastS/FastS/PyTree.py:2030-2066

The time step itself runs in libfasts_hip.so (HIP kernels, include/fasts_hip.h); nothing is
computed in Python and there is no CPU fallback.  Device copies of the fields are kept
resident; after ``_compute`` the host arrays of the current time level are refreshed unless
the environment variable FASTS_HIP_LAZY_SYNC=1 is set (then call ``sync_to_host``).
"""
import os
import numpy

from . import FastC
from . import _capi
from .FastC import Internal

__all__ = ['warmup', '_compute', '_applyBC', '_fillGhostcells', 'allocate_metric', '_init_metric',
           'display_temporal_criteria', '_createConvergenceHistory', '_computeEnstrophy', 'sync_to_host', 'sync_to_device']


def _lazy():
    return os.environ.get('FASTS_HIP_LAZY_SYNC', '0') == '1'


def _ctx():
    if FastC.HOOK is None or FastC.HOOK.get('ctx') is None:
        raise _capi.FastsHipError("fast_amd: call warmup(t, tc) before computing")
    return FastC.HOOK['ctx']


def allocate_metric(t):
    """Per-zone metrics lists (Fast/FastS/FastS/PyTree.py:272-295)."""
    own = Internal.getNodeFromName1(t, '.Solver#ownData')
    dtloc = Internal.getNodeFromName1(own, '.Solver#dtloc')[1]
    nssiter = int(dtloc[0])
    return [FastC.allocate_metric_zone(z, nssiter) for z in Internal.getZones(t)]


def _init_metric(t, metrics, hook):
    """Kept for API compatibility: Cartesian metrics are produced by allocate_metric here
    (Fast/FastS/FastS/PyTree.py:314-326 calls fastc.init_metric in the reference)."""
    return None


def joins_from_tc(t, tc):
    """Match joins stored in the connectivity tree, 'inverse' storage: donor zone carries a
    ZoneSubRegion_t 'ID_<receiver>' with PointList (donor cells) and PointListDonor (receiver
    cells), 0-based ghost-inclusive linear indices (Fast/FastS/test/tgvMpi.py:132-195).
    Returns [(rcv_zone_name, donor_zone_name, rcv_pts, donor_pts)]."""
    out = []
    if tc is None:
        return out
    for zd in Internal.getZones(tc):
        for sr in Internal.getNodesFromType1(zd, 'ZoneSubRegion_t'):
            if not sr[0].startswith('ID_'):
                continue
            rcv_name = Internal.getValue(sr)
            pl = Internal.getNodeFromName1(sr, 'PointList')
            pld = Internal.getNodeFromName1(sr, 'PointListDonor')
            if pl is None or pld is None:
                raise ValueError("warmup: join %s of zone %s lacks PointList/PointListDonor" % (sr[0], zd[0]))
            typ = Internal.getNodeFromName1(sr, 'InterpolantsType')
            if typ is not None and numpy.any(numpy.asarray(typ[1]) > 1):
                raise ValueError("warmup: only match ('direct') joins are supported (%s/%s)" % (zd[0], sr[0]))
            out.append((rcv_name, zd[0], numpy.asarray(pld[1], dtype=numpy.int32).ravel(),
                        numpy.asarray(pl[1], dtype=numpy.int32).ravel()))
    return out


def flat_tc(tc):
    """(Parameter_int, Parameter_real) of the connectivity tree when it carries the flat arrays
    (Fast/FastS/FastS/PyTree.py:486-489), else None."""
    if tc is None:
        return None
    pi = Internal.getNodeFromName1(tc, 'Parameter_int')
    pr = Internal.getNodeFromName1(tc, 'Parameter_real')
    if pi is None or pr is None or pi[1] is None or pr[1] is None:
        return None
    return (pi[1], pr[1])


def _warmup_host(t, tc, Padding=None, device=None):
    """Host part of warmup: solver data, metrics, BC table, three compacted time levels (in page-locked memory of the
    HIP runtime when a device is named and has some to give: the default refresh of the host copy after every
    _compute then runs at the rate of the host link)."""
    FastC.LEVEL_ALLOC = None if device is None else (lambda n: _capi.pinned_empty(n, device))
    FastC._reorder(t, tc)
    FastC._buildOwnData(t, Padding)
    own = Internal.getNodeFromName1(t, '.Solver#ownData')
    dtloc = Internal.getNodeFromName1(own, '.Solver#dtloc')[1]
    zones = Internal.getZones(t)
    FastC._build_omp(t)
    metrics = allocate_metric(t)
    FastC._BCcompact(t)
    FastC.fill_task_records(dtloc, zones)
    FastC.FIRST_IT = FastC._createPrimVars(t)
    return zones, metrics, dtloc


def _new_context(zones, metrics, dtloc, device):
    """Device context with every zone of t registered; returns (ctx, {zone name: index})."""
    if FastC.HOOK is not None and FastC.HOOK.get('ctx') is not None:
        FastC.HOOK['ctx'].close()
    ctx = _capi.Context(device)
    ctx.set_verif(int(dtloc[1]))
    ctx.lssiter_loc = FastC.lssiter_loc(zones)
    ctx.set_local_newton(ctx.lssiter_loc)
    names = {}
    for nd, (z, m) in enumerate(zip(zones, metrics)):
        o = Internal.getNodeFromName1(z, '.Solver#ownData')
        pint = Internal.getNodeFromName1(o, 'Parameter_int')[1]
        preal = Internal.getNodeFromName1(o, 'Parameter_real')[1]
        ctx.zone_add(pint, preal, FastC.level_buffer(z, ''), FastC.level_buffer(z, '_M1'),
                     FastC.level_buffer(z, '_P1'), m[0])
        if int(pint[27]) == 3:      # Spalart-Allmaras: wall distance (the reference compacts it behind ViscosityEddy)
            sol = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
            dist = Internal.getNodeFromName1(sol, 'TurbulentDistance')
            if dist is None:
                raise ValueError("warmup: zone %s uses the Spalart-Allmaras model and has no TurbulentDistance field" % z[0])
            ctx.zone_set_distance(nd, dist[1])
        if int(pint[62]) == 1:     # selective frequency damping: the compacted *_f fields
            ctx.zone_set_filtered(nd, FastC.level_buffer(z, '_f'))
        if int(pint[29]) == 1 and FastC.OMP_NUM_THREADS > 1:
            # LU-SGS solve in the blocks of that many OpenMP threads of the reference (distributeThreads.cpp:90-95)
            ctx.zone_set_lu_blocks(nd, FastC.lu_thread_blocks(pint, FastC.OMP_NUM_THREADS))
        names[z[0]] = nd
    return ctx, names


def _set_hook(ctx, dtloc, names, mpi=0):
    FastC.HOOK = {'ctx': ctx, 'dtloc': dtloc, 'FIRST_IT': FastC.FIRST_IT, 'param_int_tc': None,
                  'param_real_tc': None, 'mpi': mpi, 'zone_index': names,
                  'MX_SSZONE': FastC.MX_SSZONE, 'MX_SYNCHRO': FastC.MX_SYNCHRO,
                  'MX_OMP_SIZE_INT': FastC.MX_OMP_SIZE_INT, 'lssiter_loc': getattr(ctx, 'lssiter_loc', 0), 'neq_max': 5}
    return FastC.HOOK


def warmup(t, tc, graph=None, infos_ale=None, Adjoint=False, tmy=None, list_graph=None, Padding=None,
           verbose=0, device=0):
    """Perform necessary operations for the solver to run (same signature as the reference plus
    the HIP device index)."""
    if infos_ale is not None or Adjoint or tmy is not None:
        raise ValueError("fast_amd.warmup: ALE, adjoint and running statistics are not supported")
    zones, metrics, dtloc = _warmup_host(t, tc, Padding, device)
    ctx, names = _new_context(zones, metrics, dtloc, device)
    ctx.set_first_it(FastC.FIRST_IT)     # 1 when the tree carries stored _M1 levels (restart): BDF2 from the first step
    flat = flat_tc(tc)
    if flat is not None:
        # flattened connectivity (what the reference's transfer routines read): every join in one native call
        ctx.joins_set(flat[0], flat[1], 0)
    else:
        for (rn, dn, rpts, dpts) in joins_from_tc(t, tc):
            if rn not in names or dn not in names:
                raise ValueError("warmup: join %s <- %s refers to a zone that is not in t "
                                 "(zones on other ranks need fast_amd.Mpi.warmup)" % (rn, dn))
            ctx.join_add(names[rn], names[dn], rpts, dpts)
    ctx.upload()
    hook = _set_hook(ctx, dtloc, names)
    if flat is not None:
        hook['param_int_tc'], hook['param_real_tc'] = flat
    # ghost cells of the initial field: joins, BCs, edges (Fast/FastS/FastS/PyTree.py:520-527)
    ctx.fill_ghostcells(_capi.FS_LEVEL_N)
    ctx.download(_capi.FS_LEVEL_N)
    return (t, tc, metrics)


def _compute(t, metrics, nitrun, tc=None, graph=None, tc2=None, graph2=None, layer="c", NIT=1,
             ucData=None, vtune=None):
    """Compute a given number of iterations."""
    if tc2 is not None or ucData is not None:
        raise ValueError("fast_amd._compute: tc2 / ucData are not supported")
    ctx = _ctx()
    own = Internal.getNodeFromName1(t, '.Solver#ownData')
    dtloc = Internal.getNodeFromName1(own, '.Solver#dtloc')[1]
    nitmax = int(dtloc[0])
    zones = Internal.getZones(t)
    itypcp = int(Internal.getNodeFromName2(zones[0], 'Parameter_int')[1][29])
    implicit = itypcp < 2
    if layer == "Python":
        # one sub-iteration per native call, transfers and BCs driven from here
        # (Fast/FastS/FastS/PyTree.py:119-209); the last slot of an implicit step is the
        # verification pass, skipped like there (:123)
        for nstep in range(1, nitmax + 1):
            ctx.stage(nitrun, nstep)      # the last slot of an implicit step only evaluates residual norms, on verification steps
            if implicit and nstep == nitmax:
                continue
            ctx.local_joins(nstep)
            ctx.bcs(nstep)
        if implicit or nitmax % 2 == 1:
            ctx.rotate()
        nit = 1
    else:
        ctx.compute(nitrun, 1, nitmax, NIT)
        nit = NIT
    # switch pointers at the end of the time step (Fast/FastS/FastS/PyTree.py:231-233)
    case = nit % 3
    if case != 0:
        FastC.switchPointers__(zones, case, 3 if implicit else nitmax % 2 + 2)
    FastC.HOOK["FIRST_IT"] = 1
    FastC.FIRST_IT = 1
    if not _lazy():
        ctx.download(_capi.FS_LEVEL_N)
    if vtune is None:
        return None
    return 0., 0.


def _computeEnstrophy(t, metrics):
    """(enstrophy, kinetic energy) per unit volume of the current state (Fast/FastS/FastS/PyTree.py:1869-1882),
    reduced on the device."""
    s = _ctx().tgv_sums(_capi.FS_LEVEL_N)
    return (s[0] / s[2], s[1] / s[2])


def sync_to_host(t=None, levels=_capi.FS_LEVEL_N | _capi.FS_LEVEL_M1 | _capi.FS_LEVEL_P1 | _capi.FS_LEVEL_FILTERED):
    """Refresh the host arrays of the pyTree from the device (time levels and, with selective frequency damping,
    the filtered state)."""
    _ctx().download(levels)


def sync_to_device(t=None):
    """Push the host arrays of the pyTree to the device (after editing fields in Python)."""
    _ctx().upload()


def _level_of(var):
    if var.endswith('_P1'):
        return _capi.FS_LEVEL_P1
    if var.endswith('_M1'):
        return _capi.FS_LEVEL_M1
    return _capi.FS_LEVEL_N


def _applyBC(t, metrics, hook1=None, nstep=1, var="Density"):
    """Apply the BCs (+ edge/corner fix) on the time level named by var."""
    ctx = _ctx()
    ctx.apply_bc(nstep, _level_of(var))
    if not _lazy():
        ctx.download(_level_of(var))
    return None


def _fillGhostcells(zones, tc, metrics, timelevel_target, vars, nstep, omp_mode, hook1, nitmax=1,
                    rk=1, exploc=0, num_passage=1, gradP=False, TBLE=False, isWireModel=False):
    """Joins then BCs on the time level named by vars[0]."""
    ctx = _ctx()
    ctx.fill_ghostcells(_level_of(vars[0]))
    if not _lazy():
        ctx.download(_level_of(vars[0]))
    return None


def _createConvergenceHistory(t, nrecs):
    """Create the nodes that store the zonal convergence histories (Fast/FastS/FastS/PyTree.py:2143-2181):
    per zone a ZoneConvergenceHistory node (value = number of records held) with IterationNumber(nrecs) and
    RSD_L2, RSD_oo, RSD_L2_diff, RSD_oo_diff (nrecs * neq each)."""
    for z in Internal.getZones(t):
        o = Internal.getNodeFromName1(z, '.Solver#ownData')
        neq = 5
        if o is not None and Internal.getNodeFromName1(o, 'Parameter_int') is not None:
            neq = int(Internal.getNodeFromName1(o, 'Parameter_int')[1][36])
        else:
            m = Internal.getNodeFromName2(t, 'GoverningEquations')
            if m is not None and Internal.getValue(m) in ('NSTurbulent', 'nsspalart'):
                neq = 6
        zch = Internal.createUniqueChild(z, 'ZoneConvergenceHistory', 'ConvergenceHistory_t',
                                         value=numpy.zeros(1, dtype=numpy.int32))
        Internal.createUniqueChild(zch, 'IterationNumber', 'DataArray_t', numpy.zeros(nrecs, dtype=numpy.int32))
        for var in ('RSD_L2', 'RSD_L2_diff', 'RSD_oo', 'RSD_oo_diff'):
            Internal.createUniqueChild(zch, var, 'DataArray_t', numpy.zeros(nrecs * neq, dtype=numpy.float64))
    return None


def _log10_floor(v):
    # dpssiter.for:25,70-79: LOG10(max(rdm, cut0x)) with cut0x = 1e-30
    return numpy.log10(numpy.maximum(v, 1.e-30))


def residual_table_lines(zname, nitrun, nssiter, implicit, hl2, hloo, lft=1, iverb=0):
    """The lines dpssiter.for prints for one zone (Fast/FastS/FastS/DISPLAY/dpssiter.for:60-245): hl2 / hloo =
    (L2, Loo)[sub-iteration][equation].  log10 of the first sub-iteration, of every sub-iteration when iverb > 1 or
    when the largest residual grew between the first and the last (:32-36), then the drop from the first to the
    last and a warning where it is a rise.  lft 0: long format (f14.9), 1 or 2: short (f7.2)."""
    last = hl2.shape[0]
    l2log, loolog = _log10_floor(hl2), _log10_floor(hloo)
    fmt = '%14.9f' if lft == 0 else '%7.2f'
    every = iverb > 1 or bool(numpy.any(hloo[last - 1] > hloo[0]))
    lines = []
    for i in range(last):
        if i == 0 or every:       # formats 3010-4311
            it_real = i + 1 if i + 1 < last or not implicit else nssiter
            lines.append(" Res_L2 (zone=%s, ssiter=%3d %3d)" % (zname, it_real, i + 1) + ''.join('  ' + fmt % v for v in l2log[i]))
            lines.append(" Res_Loo(zone=%s, ssiter=%3d %3d)" % (zname, it_real, i + 1) + ''.join('  ' + fmt % v for v in loolog[i]))
    if last > 1 or implicit:      # formats 3020-4321 (the reference prints them at i = last_it, whatever last_it is)
        d2, doo = l2log[last - 1] - l2log[0], loolog[last - 1] - loolog[0]
        lines.append(" Res_L2 (zone=%s, Diff Ssi 1-%3d)" % (zname, last) + ''.join('  ' + fmt % v for v in d2))
        lines.append(" Res_Loo(zone=%s, Diff Ssi 1-%3d)" % (zname, last) + ''.join('  ' + fmt % v for v in doo))
        for nm, d in (('L2 ', d2), ('Loo', doo)):
            if d.max() > 0:
                # the literal of dpssiter.for:176-178 runs over a continuation line: its leading blanks are part of the text
                lines.append("Warning convergence" + 19 * ' ' + "%s Newton: dom=%s Pas de temps=%6d divergence=%12.5f"
                             % (nm, zname, nitrun, d.max()))
    return lines


def convergence_record(hl2, hloo):
    """What dpssiter.for:229-237 hands to conv2pytree.for:60-72 for one verification step: log10 of the norms of the
    first sub-iteration and their drop to the last, (L2, Loo, L2 drop, Loo drop)[equation]."""
    last = hl2.shape[0]
    l2log, loolog = _log10_floor(hl2), _log10_floor(hloo)
    return l2log[0], loolog[0], l2log[last - 1] - l2log[0], loolog[last - 1] - loolog[0]


def display_temporal_criteria(t, metrics, nitrun, format=None, gmres=None, verbose='firstlast',
                              stopAtNan=True):
    """CFL and convergence information of the last verification step (Fast/FastS/FastS/PyTree.py:2030-2066 ->
    Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp, dpssiter.for, conv2pytree.for).

    Implicit zones (and explicit ones with a local time step in the reference): the L2 / Loo norms of the Newton
    residual of every sub-iteration are printed as the reference prints them (log10; first sub-iteration, then the
    drop between the first and the last, every sub-iteration when verbose != 'firstlast' or when the residual grew)
    and, when the zone carries a ZoneConvergenceHistory node, recorded there.  Explicit zones: the same norm of the
    first-stage increment (the reference tabulates nothing for them).

    format: None / 'single' (short table), 'double' (long table), 'store' (tree only), 'none' (silent).
    Returns {zone name: (L2[neq], Loo[neq])} of the first sub-iteration; the full table of a zone is
    FastC.HOOK['residual_history'][zone name] = (L2, Loo)[sub-iteration][equation]."""
    ctx = _ctx()
    own = Internal.getNodeFromName1(t, '.Solver#ownData')
    nssiter = int(Internal.getNodeFromName1(own, '.Solver#dtloc')[1][0])
    lft = 1
    if format == 'double':
        lft = 0
    elif format in ('store', 'none'):
        lft = 3
    iverb = 0 if verbose == 'firstlast' else 2
    out = {}
    table = FastC.HOOK.setdefault('residual_history', {})
    for nd, z in enumerate(Internal.getZones(t)):
        o = Internal.getNodeFromName1(z, '.Solver#ownData')
        pint = Internal.getNodeFromName1(o, 'Parameter_int')[1]
        neq = int(pint[36])
        implicit = int(pint[29]) <= 1
        hl2, hloo = ctx.residual_history(nd, nssiter, neq)
        table[z[0]] = (hl2, hloo)
        out[z[0]] = (hl2[0], hloo[0])
        # CFL_minmaxmoy of the zone: filled by the compute call in the reference (computePT.cpp:610-643), here when
        # the norms are fetched, so that _compute does not have to wait for the device
        cfl = ctx.cfl(nd)
        node = Internal.getNodeFromName1(o, 'CFL_minmaxmoy') if o is not None else None
        if node is not None:
            node[1][0:3] = cfl
        if lft < 3:
            # display_ss_iteration.cpp:104
            print("CFL_MOY_MAX_MIN(Zone= %32.32s, nitrun=%d) %f    %f    %f " % (z[0], nitrun, cfl[2], cfl[0], cfl[1]))
        if lft < 3:
            for ln in residual_table_lines(z[0], nitrun, nssiter, implicit, hl2, hloo, lft, iverb):
                print(ln)
        # conv2pytree.for:60-72: one record per call: log10 of the first sub-iteration and the drop to the last
        zch = Internal.getNodeFromName1(z, 'ZoneConvergenceHistory')
        if zch is not None:
            itn = Internal.getNodeFromName1(zch, 'IterationNumber')[1]
            rec = int(numpy.asarray(zch[1]).ravel()[0])
            if rec < itn.size:
                itn[rec] = nitrun
                sl = slice(rec * neq, (rec + 1) * neq)
                r = convergence_record(hl2, hloo)
                Internal.getNodeFromName1(zch, 'RSD_L2')[1][sl] = r[0]
                Internal.getNodeFromName1(zch, 'RSD_oo')[1][sl] = r[1]
                Internal.getNodeFromName1(zch, 'RSD_L2_diff')[1][sl] = r[2]
                Internal.getNodeFromName1(zch, 'RSD_oo_diff')[1][sl] = r[3]
                numpy.asarray(zch[1]).ravel()[0] = rec + 1
        if stopAtNan and not numpy.all(numpy.isfinite(hl2)):
            raise FloatingPointError("NaN in the residual of zone %s at iteration %d" % (z[0], nitrun))
    return out


def displayTemporalCriteria(t, metrics, nitrun, format=None, gmres=None, verbose='firstlast', stopAtNan=True):
    """Alias kept by the reference (Fast/FastS/FastS/PyTree.py:2022-2028)."""
    return display_temporal_criteria(t, metrics, nitrun, format, gmres, verbose, stopAtNan)


_displayTemporalCriteria = displayTemporalCriteria
