"""Multi-rank path on CPU: world_size 2 and 4 over gloo (see mpi_worker.py)."""
import os
import socket
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, args, timeout=240, worker='mpi_worker.py'):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1',
                   GLOO_SOCKET_IFNAME='lo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker)] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world,blocks,nv", [(2, (1, 1, 2), (6, 5, 4)), (2, (2, 1, 1), (4, 6, 5)),
                                             (4, (1, 2, 2), (5, 4, 4)), (2, (1, 2, 2), (5, 4, 4)),
                                             (2, (2, 2, 2), (4, 4, 4)),
                                             (8, (1, 2, 4), (8, 4, 4))])     # the 8-GPU split of bench.py
def test_halo_exchange_matches_serial_multiblock(world, blocks, nv):
    rcs, outs = _run(world, list(blocks) + list(nv) + [3])
    assert rcs == [0] * world, "\n".join(outs)
    assert "MPI_WORKER_RESULT 1" in outs[0]
