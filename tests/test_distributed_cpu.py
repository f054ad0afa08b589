"""The N > 1 path on CPU: two processes, gloo backend, the oracle as the per-rank executor (there is no GPU here)."""
import os
import sys
import tempfile

import numpy as np
import pytest

from hyrise_amd import abi
from hyrise_amd.distributed import chunk_range

HERE = os.path.dirname(os.path.abspath(__file__))


def test_chunk_ranges_partition_all_chunks():
    for n_chunks in (0, 1, 7, 916, 229):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for rank in range(world):
                b, e = chunk_range(n_chunks, world, rank)
                assert 0 <= b <= e <= n_chunks
                covered.extend(range(b, e))
            assert covered == list(range(n_chunks))
    assert chunk_range(916, 8, 0) == (0, 115) and chunk_range(916, 8, 7) == (805, 916)


@pytest.mark.timeout(600)
def test_two_rank_aggregate_and_joins_match_single_process():
    """gloo, two processes, the oracle as the per-rank executor: chunk sharding + the exchanges of hyrise_amd/distributed.py
    (fixed-slot all-reduce, all-gather merge, broadcast-build all-gather, repartition all-to-all) reproduce the single-process
    operators.  tests/test_distributed_gpu.py runs the same workload with the HIP executor."""
    import pickle
    import torch.multiprocessing as mp
    import distributed_workload
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(distributed_workload.worker, args=(world, init_file, tmp, "oracle"), nprocs=world, join=True)
        results = [pickle.load(open(os.path.join(tmp, f"rank{r}.pkl"), "rb")) for r in range(world)]
    distributed_workload.check_results(results)


def test_unmergeable_aggregates_are_refused():
    """COUNT DISTINCT / STDDEV_SAMP have no cross-rank merge rule here: the sharded aggregate raises instead of returning zeros."""
    from hyrise_amd.distributed import _local_partials
    with pytest.raises(NotImplementedError):
        _local_partials(None, [], [(abi.AGG_COUNT_DISTINCT, None)])
    with pytest.raises(NotImplementedError):
        _local_partials(None, [], [(abi.AGG_STDDEV_SAMP, None)])
