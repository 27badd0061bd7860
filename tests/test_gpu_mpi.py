"""Multi-rank product path on one GPU: 2 ranks share GPU 0.  RCCL wants one GPU per rank, so the messages of these
jobs travel over gloo (tests/halo_gloo.py, staged through pinned host memory) through the transport hook of
fast_amd.Mpi.warmup; the RCCL transport of the library is covered on one rank by test_gpu_comm.py.  With four
blocks, every rank holds two: local joins and messages in the same sub-iteration."""
import pytest

from test_mpi_gloo import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocks,n,mode", [((1, 1, 2), 12, 'explicit'), ((2, 1, 1), 10, 'explicit'),
                                           ((1, 2, 1), 10, 'implicit'), ((1, 1, 2), 10, 'curvilinear'),
                                           ((1, 1, 2), 10, 'sa'),
                                           ((1, 2, 2), 8, 'explicit'), ((1, 2, 2), 8, 'walls'),
                                           ((1, 2, 2), 8, 'walls_implicit'), ((2, 2, 1), 8, 'implicit'),
                                           ((1, 2, 1), 24, 'implicit_threads')])
def test_two_ranks_one_gpu(blocks, n, mode):
    rcs, outs = _run(2, list(blocks) + [n, 3, mode], timeout=300, worker='mpi_gpu_worker.py')
    assert rcs == [0, 0], "\n".join(outs)
    assert "MPI_GPU_WORKER_RESULT 1" in outs[0]
