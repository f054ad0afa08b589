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


def test_string_keys_across_ranks_need_shared_names():
    """AggregateKeyNames of strings of five or more bytes are ids in order of first appearance inside ONE process; the sharded operators
    refuse them unless the names were built from a list every rank agrees on (distributed.shared_long_strings)."""
    from hyrise_amd.distributed import _check_key_names
    from hyrise_amd.string_keys import AggregateKeyNames
    local = AggregateKeyNames()
    assert local.name("N") == 2 + ord("N")          # short strings: the bytes themselves, the same everywhere
    _check_key_names([local, None])
    local.name("BUILDING")
    with pytest.raises(NotImplementedError):
        _check_key_names([local])
    rank0, rank1 = AggregateKeyNames(shared_long_strings=[b"MACHINERY", b"BUILDING"]), AggregateKeyNames(shared_long_strings=[b"MACHINERY", b"BUILDING"])
    assert rank1.name("BUILDING") == rank0.name("BUILDING") == 5_000_000_001 and rank0.name("MACHINERY") == 5_000_000_000
    _check_key_names([rank0])
