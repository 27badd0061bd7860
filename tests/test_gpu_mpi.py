"""Multi-rank product path on one GPU: 2 ranks share GPU 0, messages over gloo (staged through
pinned host memory).  On a multi-GPU node the same code runs one rank per GPU over RCCL."""
import pytest

from test_mpi_gloo import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocks,n,mode", [((1, 1, 2), 12, 'explicit'), ((2, 1, 1), 10, 'explicit'),
                                           ((1, 2, 1), 10, 'implicit'), ((1, 1, 2), 10, 'curvilinear'),
                                           ((1, 1, 2), 10, 'sa')])
def test_two_ranks_one_gpu(blocks, n, mode):
    rcs, outs = _run(2, list(blocks) + [n, 3, mode], timeout=300, worker='mpi_gpu_worker.py')
    assert rcs == [0, 0], "\n".join(outs)
    assert "MPI_GPU_WORKER_RESULT 1" in outs[0]
