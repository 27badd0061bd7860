"""RCCL transport of the library on one rank: the joins between the blocks of a job travel as messages from the rank to
itself (ncclSend / ncclRecv to self inside one group, fs_comm_setup handshake included) and must give, bit for bit,
what the same joins give as local copies."""
import numpy as np
import pytest

from fast_amd import FastC, Internal, _capi
import fast_amd.Mpi as FastSmpi

pytestmark = pytest.mark.gpu

SINGLE = {'cache_blocking_I': 100000, 'cache_blocking_J': 100000, 'cache_blocking_K': 100000}


def _run(n, blocks, model, loopback, nsteps, opts, flat=False):
    job = FastSmpi.tgv_job(n, blocks, 'senseur', model, 0, 1, 0, loopback=loopback, flat=flat, extra_numz=SINGLE, **opts)
    assert bool(FastC.HOOK['messages']) == loopback
    first = [FastC.level_buffer(z).copy() for z in job.zones]
    for it in range(1, nsteps + 1):
        FastSmpi._compute(job.t, job.metrics, it, job.tc, job.graph)
    last = [FastC.level_buffer(z).copy() for z in Internal.getZones(job.t)]
    return job, first, last


@pytest.mark.parametrize("blocks,n,model,opts", [
    ((1, 1, 2), 12, 'NSLaminar', {}),
    ((2, 2, 1), 10, 'NSLaminar', {'perturb': 0.15}),
    ((2, 1, 2), 70, 'Euler', {}),                         # blocks wider than one tile: several tiles feed a message
    ((1, 2, 1), 10, 'NSLaminar', {'time_step': 0.02, 'temporal_scheme': 'implicit', 'extra_numb': {'ss_iteration': 3}}),
    ((1, 1, 2), 10, 'NSTurbulent', {}),                   # 6 variables per cell in the messages
    ((1, 2, 1), 40, 'NSLaminar', {'perturb': 0.15}),      # curvilinear, several patches: donors' patches run first
    ((2, 1, 1), 36, 'NSTurbulent', {'perturb': 0.1}),
])
def test_self_send_matches_local_joins(blocks, n, model, opts):
    _, first_l, last_l = _run(n, blocks, model, False, 3, opts)
    job, first_m, last_m = _run(n, blocks, model, True, 3, opts)
    for a, b in zip(first_l, first_m):
        assert np.array_equal(a, b), "ghost cells of the initial field"
    for a, b in zip(last_l, last_m):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)
    # reductions over the (one) rank
    assert np.array_equal(job.ctx.allreduce([3., -4.], 'max'), [3., -4.])
    assert np.array_equal(job.ctx.allreduce([3., -4.], 'sum'), [3., -4.])
    if 'perturb' not in opts:      # the diagnostics are those of Cartesian zones
        assert np.isfinite(FastSmpi._computeEnstrophy(job.t, job.metrics)).all()
    job.ctx.close()


def test_self_send_from_the_flat_connectivity():
    """The same, the joins handed over as tc/Parameter_int / Parameter_real (fs_joins_set: no block is local)."""
    _, first_l, last_l = _run(10, (2, 1, 2), 'NSLaminar', False, 3, {})
    job, first_m, last_m = _run(10, (2, 1, 2), 'NSLaminar', True, 3, {}, flat=True)
    for a, b in zip(first_l + last_l, first_m + last_m):
        assert np.array_equal(a, b)
    job.ctx.close()


def test_transport_without_a_tensor_framework_in_the_process():
    """The library's own dlopen of librccl beside the system HIP runtime (this test process has torch loaded, which
    ships copies of both under the same sonames: a fresh process has not)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(here, 'rccl_loopback_notorch.py')], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_NOTORCH_RESULT 1' in r.stdout, r.stdout + r.stderr
    # the caller owns the standard output (bench.py prints one line of JSON there): RCCL's banner must not reach it
    assert r.stdout.strip().splitlines() == ['RCCL_NOTORCH_RESULT 1'], r.stdout


def test_compute_without_setup_is_refused():
    """Remote joins registered but no communicator: fs_compute fails with a message, it does not skip the messages."""
    from fast_amd import cases
    import fast_amd.PyTree as FastS
    t = cases.tgv((8, 8, 16), 'senseur', blocks=(1, 1, 2), only=[0], extra_numz=SINGLE)
    zones, metrics, dtloc = FastS._warmup_host(t, None)
    ctx, names = FastS._new_context(zones, metrics, dtloc, 0)
    pts = np.arange(10, dtype=np.int32)
    ctx.zone_set_global_id(0, 0)
    ctx.send_add_remote(0, 1, pts, 1, pts)
    ctx.upload()
    with pytest.raises(_capi.FastsHipError):
        ctx.compute(1, 1, 3, 1)
    with pytest.raises(_capi.FastsHipError):
        ctx.comm_setup()
    with pytest.raises(_capi.FastsHipError):
        ctx.halo_exchange(1)
    ctx.close()
