"""Config 5 (SSB star joins) on the device: the plans of hyrise_amd/ssb.py through the C ABI -- dimension scans, one JoinHash per
dimension over device-resident PosLists (reference columns in HBM, hy_gather_row_ids dereferencing), projection, AggregateHash --
against SQLite and against the same plan executed by the CPU oracle."""
import pytest
import torch

from hyrise_amd import ssb
from hyrise_amd.distributed import HipExecutor, aggregate_groups
from oracle_executor import OracleExecutor

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("query,sql", [("2.1", ssb.Q2_1_SQL), ("4.1", ssb.Q4_1_SQL)])
def test_ssb_query_on_device(device, query, sql):
    data = ssb.SsbData(scale_factor=0.05, seed=9, lineorder_rows=300_000)
    host = data.host_columns()
    ex = HipExecutor(torch.device("cuda", 0))
    columns = {name: ex.column(c) for name, c in host.items()}
    groupby, aggregates, joined = ssb.run_query(ex, columns, query)
    got = ssb.result_rows(aggregate_groups(ex, groupby, aggregates))
    oracle = OracleExecutor()
    o_groupby, o_aggregates, o_joined = ssb.run_query(oracle, host, query)
    want = ssb.result_rows(aggregate_groups(oracle, o_groupby, o_aggregates))
    assert joined == o_joined and got == want
    if query == "2.1":
        sqlite = sorted(((year, brand), total) for total, year, brand in data.sqlite_result(sql))
    else:
        sqlite = sorted(((year, nation), profit) for year, nation, profit in data.sqlite_result(sql))
    assert got == sqlite
    # the same query as ONE call of the library (hy_star_join_aggregate, csrc/plan.hip): the rows and the size of the join result
    from hyrise_amd.operators import star_join_aggregate
    dimensions, star_groupby, star_aggregates = ssb.star_plan(columns, query)
    result, star_joined = star_join_aggregate(dimensions, star_groupby, star_aggregates)
    assert star_joined == joined and ssb.result_rows(ssb.star_groups(result, len(star_groupby))) == sqlite


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_replicated_and_repartitioned_plans(device):
    """Two processes share the test box's GPU (gloo carries the exchanges through host memory; on a multi-GPU node the same code runs one
    rank per GPU over RCCL): lineorder chunk-sharded, Q2.1 / Q4.1 with the dimensions replicated and with customer / part joined by hash
    repartition (hy_repartition_pack, hy_join_hash over the received tuples, hy_gather_row_ids) -- SQLite's rows on every rank."""
    import os
    import pickle
    import tempfile
    import torch.multiprocessing as mp
    import ssb_workload
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(ssb_workload.worker, args=(world, os.path.join(tmp, "init"), tmp, "hip"), nprocs=world, join=True)
        results = [pickle.load(open(os.path.join(tmp, f"rank{r}.pkl"), "rb")) for r in range(world)]
    ssb_workload.check_results(results)
