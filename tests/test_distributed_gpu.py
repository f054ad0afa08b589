"""The N > 1 path with the HIP executor: two processes share the one GPU of the test box (gloo carries the exchanges through
host memory there; on a multi-GPU node the same code runs one rank per GPU over RCCL -- bench.py --gpus N), every rank runs
hy_aggregate_hash / hy_join_hash / hy_column_export / hy_repartition_pack / hy_gather_row_ids on its chunk range."""
import os
import pickle
import tempfile

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_match_single_process(device):
    import torch.multiprocessing as mp
    import distributed_workload
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(distributed_workload.worker, args=(world, init_file, tmp, "hip"), nprocs=world, join=True)
        results = [pickle.load(open(os.path.join(tmp, f"rank{r}.pkl"), "rb")) for r in range(world)]
    distributed_workload.check_results(results)
