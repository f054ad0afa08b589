"""The N > 1 path on CPU: two processes, gloo backend, the oracle as the per-rank executor (there is no GPU here).
Checks that chunk sharding + one all-gather + the deterministic merge reproduce the single-process result."""
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


def _worker(rank, world, init_file, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import pickle
    import torch.distributed as dist
    from hyrise_amd.distributed import gather_build_column, shard_column, sharded_aggregate
    from support import build_column, join_result_multiset, oracle_aggregate, oracle_join
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    rng = np.random.default_rng(11)   # same data on every rank; each rank only touches its chunk range
    n, chunk = 40_000, 3000
    k1 = rng.integers(0, 37, n).astype(np.int32) * 1009
    k1_null = rng.random(n) < 0.02
    k2 = rng.integers(0, 3, n).astype(np.int64)
    ints = rng.integers(-500, 500, n).astype(np.int32)
    floats = (rng.random(n) * 100).astype(np.float32)
    vnull = rng.random(n) < 0.05
    g1, g2 = build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY), build_column(k2, None, chunk, abi.ENC_UNENCODED)
    ci, cf = build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE), build_column(floats, None, chunk, abi.ENC_DICTIONARY)
    aggregates = [(abi.AGG_SUM, ci), (abi.AGG_AVG, cf), (abi.AGG_MIN, ci), (abi.AGG_MAX, cf), (abi.AGG_COUNT, ci), (abi.AGG_COUNT, None)]
    rows = sharded_aggregate(dist, oracle_aggregate, [g1, g2], aggregates)

    # join: broadcast-build -- every rank holds a shard of the build column's values, all-gathers them, probes its shard
    build_values = rng.integers(0, 5000, 9000).astype(np.int32)
    probe_values = rng.integers(0, 6000, 50_000).astype(np.int32)
    per = (len(build_values) + world - 1) // world
    full_build, _ = gather_build_column(dist, build_values[rank * per:(rank + 1) * per], None)
    assert np.array_equal(full_build, build_values)
    probe_full = build_column(probe_values, None, 4096, abi.ENC_UNENCODED)
    probe_shard, chunk_begin = shard_column(probe_full, world, rank)
    build_col = build_column(full_build, None, 2000, abi.ENC_UNENCODED)
    local = oracle_join(build_col, probe_shard, abi.JOIN_INNER)
    pairs = [(l, (r[0] + chunk_begin, r[1])) for l, r in join_result_multiset(local, abi.JOIN_INNER)]
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as fh:
        pickle.dump({"aggregate": rows, "pairs": pairs}, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_aggregate_and_join_match_single_process():
    import pickle
    import torch.multiprocessing as mp
    from support import build_column, join_result_multiset, oracle_aggregate, oracle_join, column_values
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(world, init_file, tmp), nprocs=world, join=True)
        results = [pickle.load(open(os.path.join(tmp, f"rank{r}.pkl"), "rb")) for r in range(world)]
    # single-process reference
    rng = np.random.default_rng(11)
    n, chunk = 40_000, 3000
    k1 = rng.integers(0, 37, n).astype(np.int32) * 1009
    k1_null = rng.random(n) < 0.02
    k2 = rng.integers(0, 3, n).astype(np.int64)
    ints = rng.integers(-500, 500, n).astype(np.int32)
    floats = (rng.random(n) * 100).astype(np.float32)
    vnull = rng.random(n) < 0.05
    g1, g2 = build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY), build_column(k2, None, chunk, abi.ENC_UNENCODED)
    ci, cf = build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE), build_column(floats, None, chunk, abi.ENC_DICTIONARY)
    aggregates = [(abi.AGG_SUM, ci), (abi.AGG_AVG, cf), (abi.AGG_MIN, ci), (abi.AGG_MAX, cf), (abi.AGG_COUNT, ci), (abi.AGG_COUNT, None)]
    want = oracle_aggregate([g1, g2], aggregates)
    g1_values, g2_values = column_values(g1), column_values(g2)
    flat = {}
    offset = 0
    for c, seg in enumerate(g1.segments):
        for i in range(seg.size):
            flat[(c, i)] = offset + i
        offset += seg.size
    for rank_rows in (results[0]["aggregate"], results[1]["aggregate"]):   # every rank ends with the same merged result
        assert len(rank_rows) == want.n_groups
        for g, (first, row) in enumerate(rank_rows):
            rid = tuple(int(x) for x in want.row_ids[g])
            assert first == rid, "group order = first occurrence over the whole table"
            assert row[0] == g1_values[flat[rid]] and row[1] == g2_values[flat[rid]]
            for a in range(len(aggregates)):
                expected = want.column(a)[g]
                got = row[2 + a]
                if expected is None:
                    assert got is None
                elif isinstance(expected, float):
                    assert abs(got - expected) <= 1e-9 * max(1.0, abs(expected))
                else:
                    assert got == expected
    build_values = rng.integers(0, 5000, 9000).astype(np.int32)
    probe_values = rng.integers(0, 6000, 50_000).astype(np.int32)
    full = oracle_join(build_column(build_values, None, 2000, abi.ENC_UNENCODED), build_column(probe_values, None, 4096, abi.ENC_UNENCODED), abi.JOIN_INNER)
    want_pairs = join_result_multiset(full, abi.JOIN_INNER)
    got_pairs = sorted(results[0]["pairs"] + results[1]["pairs"], key=lambda p: (p[0], p[1]))
    assert got_pairs == sorted(want_pairs, key=lambda p: (p[0], p[1]))
