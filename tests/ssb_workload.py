"""The two-rank SSB workload (config 5 at N > 1): lineorder chunk-sharded, the dimensions on every rank; Q2.1 and Q4.1 with every
dimension joined against its full replica, and with `customer` and `part` joined by HASH REPARTITION (hyrise_amd/ssb.py
_join_dimension_repartitioned: lineorder tuples travel to the rank that owns the key and back), the groups combined by the sharded
AggregateHash.  The executor decides where the per-rank work runs: the CPU oracle (tests/test_ssb_cpu.py) or the HIP library on a
GPU both ranks share (tests/test_ssb_gpu.py) -- the exchange code is the same."""
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SCALE, SEED, LINEORDER_ROWS, CHUNK = 0.02, 5, 60_000, 7000
PLANS = {"replicated": (), "repartitioned": ("customer", "part")}


def worker(rank, world, init_file, out_dir, executor_kind):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from hyrise_amd import abi, ssb
    from hyrise_amd.distributed import Comm, shard_column, sharded_aggregate
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    comm = Comm(dist).bind(torch.device("cpu"))
    if executor_kind == "hip":
        from hyrise_amd.distributed import HipExecutor
        lib = abi.load_library()
        abi.check(lib.hy_init(0))
        ex = HipExecutor(torch.device("cuda", 0))
    else:
        from oracle_executor import OracleExecutor
        ex = OracleExecutor()
    data = ssb.SsbData(scale_factor=SCALE, seed=SEED, lineorder_rows=LINEORDER_ROWS)
    host = data.host_columns(chunk_size=CHUNK)
    fact = set(ssb.SsbData.TABLES["lineorder"])
    columns, first_chunk = {}, 0
    for name, column in host.items():
        if name in fact:
            column, first_chunk = shard_column(column, world, rank)
        columns[name] = ex.column(column)
    out = {}
    for query in ("2.1", "4.1"):
        for plan, repartitioned in PLANS.items():
            groupby, aggregates, joined = ssb.run_query(ex, columns, query, comm=comm, repartitioned=repartitioned)
            groups = sharded_aggregate(comm, ex, groupby, aggregates, first_chunk)
            out[(query, plan)] = (ssb.result_rows(groups), joined)
        if executor_kind == "hip":   # the ranks' shards through hy_star_join_aggregate, partial groups added up over the ranks
            groups, joined, _ = ssb.sharded_star_groups(comm, columns, query)
            out[(query, "one call per rank")] = (ssb.result_rows(groups), joined)
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def check_results(results):
    """Every rank holds the whole result of every (query, plan); it is SQLite's; the ranks' joined rows add up to the query's."""
    from hyrise_amd import ssb
    data = ssb.SsbData(scale_factor=SCALE, seed=SEED, lineorder_rows=LINEORDER_ROWS)
    want = {"2.1": sorted(((year, brand), total) for total, year, brand in data.sqlite_result(ssb.Q2_1_SQL)),
            "4.1": sorted(((year, nation), profit) for year, nation, profit in data.sqlite_result(ssb.Q4_1_SQL))}
    for query in ("2.1", "4.1"):
        joined_replicated = sum(result[(query, "replicated")][1] for result in results)
        assert joined_replicated > 0
        for plan in list(PLANS) + (["one call per rank"] if (query, "one call per rank") in results[0] else []):
            assert sum(result[(query, plan)][1] for result in results) == joined_replicated, f"Q{query} {plan}: joined rows"
            for rank, result in enumerate(results):
                assert result[(query, plan)][0] == want[query], f"Q{query}, {plan} plan, rank {rank}: groups differ from SQLite's"
