#!/usr/bin/env python3
"""Headline benchmark: Mcell-iter/s of the fp64 RK3 Taylor-Green step on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment)

Workload (N = 1): Taylor-Green vortex Re = 1600, M = 0.1, one Cartesian block of 256^3 cells
(+2 ghost layers), NSLaminar, scheme 'senseur' (the scheme of Fast/FastS/test/tgvMpi.py:272),
explicit RK3, 6 x BCautoperiod, synthetic deterministic initial field.  One step = one full
time step = 3 sub-iterations (flux balance + RK update + ghost fill each).  Fields are
resident in HBM before the timed region; nothing is copied to the host inside it.
N > 1: weak scaling, one 256^3 block per GPU of a (bx,by,bz)-split periodic box, match joins
exchanged with RCCL send/recv inside libfasts_hip.so (fs_comm_*) between the ranks.

Prints ONE JSON line (rank 0) with the whole-job throughput, the roofline of the dominant
kernel (RK-stage kernel; HIP events on the launch stream) and a CPU baseline measured on a
bounded sample: the compiled reference on all host cores of the job (one thread beside it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
BYTES_PER_CELL_STEP = {'senseur': 472.0, 'ausmpred': 400.0, 'roe': 400.0}


def split_for(n):
    """Blocks of the periodic box per direction.  The i direction is never cut: a tile of the stage kernel is 64 cells
    wide along i, so a cut there would put half of a 256-cell block into the tiles that feed halo messages."""
    return {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (1, 2, 4)}.get(n)


def block_cells(n, world):
    """Cells of one block (one per GPU, n^3 cells each).  8 GPUs: the 512^3 box of the scaling configuration cut in
    1 x 2 x 4 blocks of 2n x n x n/2 cells; fewer GPUs: cubic blocks."""
    return (2 * n, n, n // 2) if world == 8 and n % 2 == 0 else (n, n, n)


def _cpu_case(n, scheme):
    from fast_amd import cases, FastC, Internal
    t = cases.tgv(n, scheme)
    FastC._buildOwnData(t, None); FastC._build_omp(t)
    dtloc = Internal.getNodeFromName2(t, '.Solver#dtloc')[1]
    z = Internal.getZones(t)[0]
    metrics = FastC.allocate_metric_zone(z, int(dtloc[0]))
    FastC._BCcompact(t); FastC._createPrimVars(t)
    pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
    preal = Internal.getNodeFromName2(z, 'Parameter_real')[1]
    return z, pint, preal, FastC.level_buffer(z), metrics


def _timed(step, n, budget_s):
    step(1)
    t0 = time.time(); it = 0
    while True:
        it += 1
        step(it + 1)
        el = time.time() - t0
        if (el > budget_s and it >= 2) or it >= 50:
            break
    return n ** 3 * it / el / 1e6, it


def _ref_solver(n, scheme, threads):
    from fast_amd import Internal
    from oracle import ref_harness as R
    z, pint, preal, ro, metrics = _cpu_case(n, scheme)
    g = Internal.getNodeFromName1(z, 'GridCoordinates')
    xyz = [Internal.getNodeFromName1(g, c)[1] for c in ('CoordinateX', 'CoordinateY', 'CoordinateZ')]
    zn = R.RefZone(pint, preal, xyz[0], xyz[1], xyz[2], ro, metrics[0])
    s = R.RefSolver([zn]); s.threads = threads; s.fill_ghostcells('ro')
    return s


def cpu_baseline(scheme, budget_s=10.0):
    """CPU legs on bounded samples of the same workload (the same TGV case, smaller box).
    value = the compiled reference Fortran on every host core given to the job: its own thread decomposition
            (indice_boucle_lu) and lock protocol, driven like its parallel region (oracle/ref_mt.c); BCs on one thread.
    single_thread = the same on one thread.  Without the compiled reference: the C restatement on OpenMP threads."""
    from oracle import oracle as O
    from oracle import ref_harness as R
    cores = max(1, min(int(os.environ.get('OMP_NUM_THREADS', '0') or 0) or (os.cpu_count() or 1), os.cpu_count() or 1))
    n = 160
    if R.available():
        s = _ref_solver(n, scheme, cores)
        v, it = _timed(lambda i: s.step(i), n, budget_s)
        out = {"value": v, "unit": "Mcell-iter/s", "cores": cores, "kind": "reference",
               "sample": "TGV %d^3 %s RK3, %d steps, compiled reference Fortran on %d OpenMP threads, default cache blocking"
                         % (n, scheme, it, cores)}
        n1 = 96
        s1 = _ref_solver(n1, scheme, 1)
        v1, it1 = _timed(lambda i: s1.step(i), n1, budget_s * 0.6)
        out["single_thread"] = {"value": v1, "unit": "Mcell-iter/s", "cores": 1, "kind": "reference",
                                "sample": "TGV %d^3 %s RK3, %d steps, one thread" % (n1, scheme, it1)}
        return out
    z, pint, preal, ro, metrics = _cpu_case(n, scheme)
    oz = O.OracleZone(pint, preal, ro, metrics[0]); oz.fill_ghostcells()
    oz.threads = cores
    v, it = _timed(lambda i: oz.step(), n, budget_s)
    return {"value": v, "unit": "Mcell-iter/s", "cores": cores, "kind": "port",
            "sample": "TGV %d^3 %s RK3, %d steps, C restatement on %d OpenMP threads (k-slabs)" % (n, scheme, it, cores)}


def visible_gpus():
    """GPUs a HIP process started from here would see, counted in a child process: the launcher itself never touches
    the GPU (it forks)."""
    import subprocess
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "rc = ctypes.CDLL('libamdhip64.so').hipGetDeviceCount(ctypes.byref(n))\n"
            "print(n.value if rc == 0 else 0)\n")
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(n, argv, ngpu=None, command=None, timeout=3000.):
    """`python bench.py --gpus N` with no launcher around it: start N fresh processes of this script, one per GPU (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR / MASTER_PORT and a communicator-id file of their own in the environment), pass on
    what rank 0 prints, and return 0 only if every rank did.  Fewer than N visible GPUs: an error, never a smaller job.
    command: what to start instead of this script (the test of the launcher)."""
    import socket
    import subprocess
    import tempfile
    if ngpu is None:
        ngpu = visible_gpus()
    if ngpu < n:
        print("bench.py: --gpus %d but %d GPU(s) visible" % (n, ngpu), file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    idfile = os.path.join(tempfile.mkdtemp(prefix='fasts_hip_'), 'comm_id')
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), FASTS_HIP_ID_FILE=idfile, HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = (command or [sys.executable, os.path.abspath(__file__)]) + list(argv)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    t0 = time.time()
    rc, out0 = 0, None
    try:
        while True:
            if out0 is None:
                try:
                    out0, _ = procs[0].communicate(timeout=0.5)
                except subprocess.TimeoutExpired:
                    pass
            else:
                time.sleep(0.2)
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:          # one rank down: the others would wait for it for ever
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > timeout:
                rc = 124
                break
    finally:
        for p in procs:          # the ranks this call started, by their own process ids
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    try:
        os.remove(idfile)
    except OSError:
        pass
    try:
        os.rmdir(os.path.dirname(idfile))
    except OSError:
        pass
    if rc == 0:
        sys.stdout.write(out0 or '')
        sys.stdout.flush()
    else:
        print("bench.py: a rank of the %d-GPU job failed (exit status %s)" % (n, rc), file=sys.stderr)
    return rc if rc else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--cells', dest='n', type=int, default=256, help='cells per direction per GPU block')
    ap.add_argument('--scheme', default='senseur', choices=['senseur', 'ausmpred', 'roe'])
    ap.add_argument('--slope', default='o3', choices=['o3', 'o3sc'], help="o3sc: the o3 states weighted by a shock sensor "
                    "(kernels of their own, one thread per cell); single GPU, TGV case")
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--sa', action='store_true', help='Spalart-Allmaras on the Cartesian box between two no-slip walls (j faces); '
                    'single GPU; not the headline workload')
    ap.add_argument('--implicit', type=int, default=0, metavar='SS_ITERATION', help='implicit BDF/LU-SGS steps with '
                    'this many Newton sub-iterations; not the headline workload')
    ap.add_argument('--perturb', type=float, default=0., help='node displacement in mesh steps: > 0 makes the block '
                    'curvilinear (per-cell metrics); not the headline workload')
    ap.add_argument('--naca', type=int, default=0, metavar='NFOIL', help='steady-RANS set-up on a two-dimensional NACA0012 '
                    'C-mesh (Spalart-Allmaras, implicit LU-SGS, local time step; 2*NFOIL + 2*(NFOIL/3) cells around, '
                    '2*NFOIL/3 across; Newton sub-iterations from --implicit, default 3); not the headline workload')
    ap.add_argument('--lu-threads', type=int, default=1, metavar='T', help='with --implicit / --naca: LU-SGS solve in the '
                    'blocks of T OpenMP threads of the reference (its result with OMP_NUM_THREADS=T); 1 = one sweep per zone')
    ap.add_argument('--nb-relax', type=int, default=1, metavar='R', help='with --implicit / --naca: LU-SSOR with R relaxations')
    ap.add_argument('--local-newton', type=int, default=0, metavar='S', help="with --implicit: temporal_scheme "
                    "'implicit_local', Newton sub-iterations on the sub-zones (blocks of S^3 cells) that have not converged; "
                    "modulo_verif 7, the reference's default for this scheme")
    ap.add_argument('--epsi-newton', type=float, default=0.1, help='convergence threshold of the Newton sub-iterations '
                    '(--local-newton)')
    ap.add_argument('--cfl', type=float, default=1., help='CFL number of the local time step (--naca)')
    ap.add_argument('--channel', type=int, nargs=3, default=None, metavar=('BI', 'BJ', 'BK'), help='one GPU\'s share of a '
                    'multi-block channel: the --cells^3 box cut in BI x 1 x BK curvilinear blocks (BJ must be 1), no-slip '
                    'walls on jmin / jmax, periodic along i and k through match joins; not the headline workload')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: be the launcher (before anything touches the GPU)
        if split_for(a.gpus) is None:
            raise SystemExit("bench.py: --gpus must be 1, 2, 4 or 8")
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    os.environ['FASTS_HIP_LAZY_SYNC'] = '1'
    from fast_amd import cases, FastC, Internal
    import fast_amd.PyTree as FastS
    import fast_amd.Mpi as FastSmpi

    model = 'Euler' if a.scheme == 'roe' else 'NSLaminar'
    if a.sa:
        if world > 1 or a.naca or a.channel or a.slope != 'o3':
            raise SystemExit("bench.py: --sa is a single-GPU option of the plain box")
        model = 'NSTurbulent'
    FastC.OMP_NUM_THREADS = max(1, a.lu_threads)

    if (a.perturb or a.implicit) and world > 1:
        raise SystemExit("bench.py: --perturb / --implicit are single-GPU options")
    if a.naca and world > 1:
        raise SystemExit("bench.py: --naca is a single-GPU option")
    if a.channel and (world > 1 or a.channel[1] != 1 or a.naca):
        raise SystemExit("bench.py: --channel BI 1 BK is a single-GPU option")
    nblk = 1
    if a.naca:
        from fast_amd import connect
        import fast_amd.pytree_lite as C
        nf, nw, nj = a.naca, a.naca // 3, 2 * a.naca // 3
        a.implicit = a.implicit or 3
        t = cases.naca_cmesh(nfoil=nf, nwake=nw, nj=nj, scheme=a.scheme, ss_iteration=a.implicit, cfl=a.cfl,
                             extra_numz={'nb_relax': a.nb_relax} if a.nb_relax > 1 else None)
        tc = C.node2Center(t)
        rp, dp = connect.cmesh_wake_join(2 * nf + 2 * nw, nj, nw)
        connect.add_joins_to_tc(tc, [(0, 0, rp, dp)], [z[0] for z in Internal.getZones(t)])
        (t, tc, metrics) = FastS.warmup(t, tc, device=local_rank)
        ctx = FastC.HOOK['ctx']
        run_step = lambda it: FastS._compute(t, metrics, it, tc)
        ncell_total = (2 * nf + 2 * nw) * nj
    elif a.channel:
        from fast_amd import connect
        import fast_amd.pytree_lite as C
        blocks = tuple(a.channel)
        nblk = blocks[0] * blocks[1] * blocks[2]
        a.perturb = a.perturb or 0.2
        nv = tuple(a.n // b for b in blocks)
        kw = {}
        if a.implicit:
            kw = {'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': a.implicit}}
        t = cases.tgv(a.n, a.scheme, model=model, time_step=0.003 * 32. / a.n * (10. if a.implicit else 1.), blocks=blocks,
                      perturb=a.perturb, bcs={'jmin': 'BCWall', 'jmax': 'BCWall'}, **kw)
        tc = C.node2Center(t)
        connect.add_joins_to_tc(tc, connect.periodic_block_joins(blocks, nv), [z[0] for z in Internal.getZones(t)])
        (t, tc, metrics) = FastS.warmup(t, tc, device=local_rank)
        ctx = FastC.HOOK['ctx']
        run_step = lambda it: FastS._compute(t, metrics, it, tc)
        ncell_total = nv[0] * nv[1] * nv[2] * nblk
    elif world == 1:
        kw = {}
        if a.implicit:
            kw = {'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': a.implicit}}
        if a.implicit and a.nb_relax > 1:
            kw['extra_numz'] = {'nb_relax': a.nb_relax}
        if a.slope != 'o3':
            kw['extra_numz'] = dict(kw.get('extra_numz') or {}, slope=a.slope)
        if a.implicit and a.local_newton:
            kw['temporal_scheme'] = 'implicit_local'
            kw['modulo_verif'] = 7
            kw['extra_numz'] = dict(kw.get('extra_numz') or {}, ssdom_IJK=[a.local_newton] * 3, epsi_newton=a.epsi_newton)
        if a.sa:
            kw['bcs'] = {'jmin': 'BCWall', 'jmax': 'BCWall'}
        t = cases.tgv(a.n, a.scheme, model=model, time_step=0.003 * 32. / a.n * (10. if a.implicit else 1.),
                      perturb=a.perturb, **kw)
        (t, tc, metrics) = FastS.warmup(t, None, device=local_rank)
        ctx = FastC.HOOK['ctx']
        nss = a.implicit + 1 if a.implicit else 3
        run_step = lambda it: ctx.compute(it, 1, nss, 1)
        ncell_total = a.n ** 3
    else:
        blocks = split_for(world)
        if blocks is None:
            raise SystemExit("bench.py: --gpus must be 1, 2, 4 or 8")
        job = FastSmpi.tgv_job(block_cells(a.n, world), blocks, a.scheme, model, rank, world, local_rank)
        ctx = job.ctx
        run_step = job.step
        ncell_total = a.n ** 3 * world

    barrier = FastSmpi.barrier     # this rank's stream drained, then an all-reduce over the ranks (RCCL)

    for it in range(1, a.warmup + 1):
        run_step(it)
    ctx.stage_kernel_ms(reset=True)      # start HIP-event timing of the stage kernel
    barrier()
    t0 = time.perf_counter()
    for it in range(a.warmup + 1, a.warmup + a.steps + 1):
        run_step(it)
    barrier()
    el = time.perf_counter() - t0
    el = float(FastSmpi.allreduce([el], 'max')[0])
    k_ms, k_n = ctx.stage_kernel_ms(reset=False)
    triad = None
    if rank == 0:
        try:
            triad = ctx.stream_triad(1 << 27, 20)
        except Exception:
            triad = None

    # sanity: the field must still be finite
    ctx.download(1)
    z0 = Internal.getZones(t)[0] if world == 1 else job.zone
    if not np.isfinite(FastC.level_buffer(z0)).all():
        raise SystemExit("bench.py: non-finite field after the run")

    local_share = None
    if a.implicit and a.local_newton and world == 1 and not a.naca and not a.channel:
        # share of the cells each Newton sub-iteration works on, from the lists the library holds now
        local_share = []
        for ns in range(1, a.implicit + 1):
            bx = ctx.subzone_list(0, ns)
            cells = sum(int((b[1] - b[0] + 1) * (b[3] - b[2] + 1) * (b[5] - b[4] + 1)) for b in bx)
            local_share.append(round(cells / float(a.n ** 3), 3))
    if rank == 0:
        # HBM bytes per launch from the PMC passes of this same command (collected in runs of their
        # own by profiles/run_profiles.sh; counters cannot be read inside a timed run)
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, 'profiles', 'traffic_%s_%d%s.json' % (a.scheme, a.n, '_curvilinear' if a.perturb else ''))
        if os.path.exists(tf) and not a.naca and world == 1 and not a.implicit and not a.channel and not a.sa and a.slope == 'o3':
            with open(tf) as fh:
                traffic = json.load(fh).get('traffic_bytes_per_launch')
            traffic_source = ("not measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of this "
                              "command, recorded in profiles/%s" % os.path.basename(tf))
        # algorithmic bytes per launch; per-cell metrics (9 surface-vector components + volume) are read once per stage
        # Spalart-Allmaras: six variables instead of five, and the wall distance and the eddy viscosity read once per stage
        per_cell = BYTES_PER_CELL_STEP[a.scheme] * (1.2 if (a.sa or a.naca) else 1.) + (48.0 if (a.sa or a.naca) else 0.)
        bpl = (per_cell + (240.0 if (a.perturb or a.naca) else 0.)) / 3.0 * (ncell_total // world // nblk)
        ach = bpl / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.
        out = {
            "metric": "Mcell-iter/s (fp64 RK3 TGV 256^3)",
            "value": ncell_total * a.steps / el / 1e6,
            "unit": "Mcell-iter/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "NACA0012 C-mesh %d x %d cells, NSTurbulent (Spalart-Allmaras), scheme=%s, wall + far field + "
                                   "wake match join" % (2 * a.naca + 2 * (a.naca // 3), 2 * a.naca // 3, a.scheme) if a.naca else
                                   "channel share: %d^3 cells in %d x %d x %d curvilinear blocks, %s, scheme=%s, RK3, no-slip walls on "
                                   "j, periodic i and k through match joins" % ((a.n,) + tuple(a.channel) + (model, a.scheme)) if a.channel else
                                   "box of %d^3 cells between two no-slip walls, NSTurbulent (Spalart-Allmaras), scheme=%s, RK3"
                                   % (a.n, a.scheme) if a.sa else
                                   "TGV Re=1600 M=0.1, %d^3 cells per GPU, NSLaminar, scheme=%s, RK3, periodic"
                                   % (a.n, a.scheme) if model != 'Euler' else
                                   "TGV %d^3 cells per GPU, Euler, scheme=%s, RK3, periodic" % (a.n, a.scheme),
                       "blocks": list(a.channel) if a.channel else list(split_for(world) or (1, 1, 1)),
                       "block_cells": [a.n // b for b in a.channel] if a.channel else list(block_cells(a.n, world)),
                       "ghost_layers": 2,
                       "slope": a.slope,
                       "mesh": "two-dimensional C-mesh" if a.naca else ("curvilinear (perturb %g)" % a.perturb if a.perturb else "cartesian"),
                       "time_scheme": ("implicit BDF1 local time step, LU-SGS, %d Newton sub-iterations" if a.naca else
                                       "implicit BDF2 LU-SGS, %d Newton sub-iterations") % a.implicit +
                                      (", %d LU-SSOR relaxations" % a.nb_relax if a.nb_relax > 1 else "") +
                                      (", LU-SGS in the blocks of %d threads" % a.lu_threads if a.lu_threads > 1 else "") +
                                      (", implicit_local: sub-zones of %d^3-cell blocks, cells worked on per sub-iteration "
                                       "at the end of the run: %s" % (a.local_newton, local_share) if local_share else "")
                                      if a.implicit else "explicit RK3"},
            "roofline": {"bound": "hbm", "kernel": "rk_stage", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": bpl, "avg_launch_ms": k_ms, "launches": k_n,
                         # what a plain streaming kernel reaches on this device, measured in this run (3 x 1 GiB triad)
                         "stream_triad": triad, "frac_of_stream_triad": ach / triad if triad else None},
        }
        if not a.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(a.scheme)
            except Exception as e:   # the baseline is informative; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mcell-iter/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)
    barrier()
    ctx.close()


if __name__ == '__main__':
    main()
