"""Config 5 (SSB star joins) on the CPU: the plans of hyrise_amd/ssb.py executed by the oracle operators reproduce what SQLite
computes from the query text (the reference's own verification practice, SQLiteTestRunner / --verify)."""
import numpy as np
import pytest

from hyrise_amd import ssb
from hyrise_amd.distributed import aggregate_groups
from oracle_executor import OracleExecutor


@pytest.fixture(scope="module")
def data():
    return ssb.SsbData(scale_factor=0.02, seed=5, lineorder_rows=60_000)


def oracle_rows(data, query):
    ex = OracleExecutor()
    columns = data.host_columns(chunk_size=7000)
    groupby, aggregates, joined = ssb.run_query(ex, columns, query)
    return ssb.result_rows(aggregate_groups(ex, groupby, aggregates)), joined


def test_q2_1_matches_sqlite(data):
    got, joined = oracle_rows(data, "2.1")
    want = sorted(((year, brand), total) for total, year, brand in data.sqlite_result(ssb.Q2_1_SQL))
    assert joined > 0 and got == want


def test_q4_1_matches_sqlite(data):
    got, joined = oracle_rows(data, "4.1")
    want = sorted(((year, nation), profit) for year, nation, profit in data.sqlite_result(ssb.Q4_1_SQL))
    assert joined > 0 and got == want


def test_table_shapes_follow_the_specification():
    d = ssb.SsbData(scale_factor=2.0, seed=1, lineorder_rows=1000)
    assert len(d.d_datekey) == 2557 and d.d_datekey[0] == 19920101 and d.d_datekey[-1] == 19981231
    assert len(d.c_custkey) == 60_000 and len(d.s_suppkey) == 4_000 and len(d.p_partkey) == 400_000
    assert set(np.unique(d.p_category // 10)) == {1, 2, 3, 4, 5} and d.p_brand1.min() >= 1101 and d.p_brand1.max() <= 5540
    assert ssb.referenced_bytes(d, "2.1") == 4 * (4 * 1000 + 2 * 400_000 + 2 * 4_000 + 2 * 2557)


@pytest.mark.timeout(600)
def test_two_ranks_replicated_and_repartitioned_plans_match_sqlite():
    """gloo, two processes, the oracle as the per-rank executor: lineorder chunk-sharded; Q2.1 / Q4.1 with the dimensions replicated and
    with `customer` / `part` joined by hash repartition (tuples to the key's rank and back) -- both SQLite's rows on every rank."""
    import os
    import pickle
    import tempfile
    import torch.multiprocessing as mp
    import ssb_workload
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(ssb_workload.worker, args=(world, os.path.join(tmp, "init"), tmp, "oracle"), nprocs=world, join=True)
        results = [pickle.load(open(os.path.join(tmp, f"rank{r}.pkl"), "rb")) for r in range(world)]
    ssb_workload.check_results(results)
