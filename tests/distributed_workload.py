"""The two-rank workload of the N > 1 tests: one table, every rank takes its chunk range, runs the sharded AggregateHash
(fixed-slot all-reduce and general all-gather merge), the broadcast-build join and the hash-repartition join, and writes
what it got.  The executor decides where the per-rank work runs: the CPU oracle (tests/test_distributed_cpu.py) or the
HIP library on a GPU both ranks share (tests/test_distributed_gpu.py) -- the exchange code is the same."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def make_data():
    from hyrise_amd import abi
    from support import build_column
    rng = np.random.default_rng(11)   # the same table on every rank; each rank only touches its chunk range
    n, chunk = 40_000, 3000
    d = {"n": n, "chunk": chunk}
    k1 = rng.integers(0, 37, n).astype(np.int32) * 1009
    k1_null = rng.random(n) < 0.02
    k2 = rng.integers(0, 3, n).astype(np.int64)
    small = rng.integers(0, 5, n).astype(np.int32)
    ints = rng.integers(-500, 500, n).astype(np.int32)
    floats = (rng.random(n) * 100).astype(np.float32)
    vnull = rng.random(n) < 0.05
    d["g1"] = build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY)
    d["g2"] = build_column(k2, None, chunk, abi.ENC_UNENCODED)
    d["g3"] = build_column(small, None, chunk, abi.ENC_DICTIONARY)
    d["gf"] = build_column((small * 0.5).astype(np.float32), None, chunk, abi.ENC_UNENCODED)
    d["ci"] = build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE)
    d["cf"] = build_column(floats, None, chunk, abi.ENC_DICTIONARY)
    d["build_values"] = rng.permutation(np.arange(0, 27000, 3, dtype=np.int32))          # unique keys, 9000 rows = 3 whole chunks of 3000
    d["probe_values"] = rng.integers(0, 30000, 50_000).astype(np.int32)
    d["build"] = build_column(d["build_values"], None, chunk, abi.ENC_UNENCODED)
    d["probe"] = build_column(d["probe_values"], None, 4096, abi.ENC_FRAME_OF_REFERENCE)
    d["nullable_probe"] = build_column(d["probe_values"][:30_000], rng.random(30_000) < 0.04, 4096, abi.ENC_DICTIONARY)   # NULL keys on the outer side
    dup_values = rng.integers(0, 2000, 9000).astype(np.int32)                           # duplicate keys on both sides
    d["dup_build"] = build_column(dup_values, rng.random(9000) < 0.03, chunk, abi.ENC_UNENCODED)
    # a small dimension whose keys are all even: with two ranks, rank 1 (key % 2 == 1) receives NO build tuples -- its local join has an
    # empty side, and the modes that emit rows without a partner must still answer (round-3 advisor finding)
    d["even_build"] = build_column((np.arange(40, dtype=np.int32) * 2), None, chunk, abi.ENC_UNENCODED)
    return d


def aggregate_specs(d):
    from hyrise_amd import abi
    full = [(abi.AGG_SUM, "ci"), (abi.AGG_AVG, "cf"), (abi.AGG_MIN, "ci"), (abi.AGG_MAX, "cf"), (abi.AGG_COUNT, "ci"), (abi.AGG_COUNT, None)]
    return {"two_keys_general": (["g1", "g2"], full),        # 37 x 1009-spaced keys: too sparse for slots -> all-gather merge
            "small_domain_slots": (["g3", "g2"], full),      # 5 x 3 keys -> fixed slots, all-reduce
            "immediate_key": (["g3"], full[:3]),             # one dense int32 key: key order, NULL first
            "float_key": (["gf"], full[:2]),                 # floating-point GROUP BY column -> general merge
            "no_groupby": ([], full),
            # COUNT(DISTINCT) and STDDEV_SAMP across ranks: the distinct (group, value) tuples all-gathered, (n, sum, M2) merged pairwise
            "distinct_and_stddev_slots": (["g3"], [(abi.AGG_COUNT_DISTINCT, "ci"), (abi.AGG_STDDEV_SAMP, "cf"), (abi.AGG_SUM, "ci"), (abi.AGG_STDDEV_SAMP, "ci")]),
            "distinct_and_stddev_general": (["g1", "g2"], [(abi.AGG_STDDEV_SAMP, "cf"), (abi.AGG_COUNT_DISTINCT, "g3"), (abi.AGG_AVG, "cf")]),
            "stddev_no_groupby": ([], [(abi.AGG_STDDEV_SAMP, "cf"), (abi.AGG_COUNT_DISTINCT, "ci")])}


def fused_specs(d):
    """Plans for the sharded fused pass: (filters [(column name, predicate)], GROUP BY names, aggregates [(function, tree over names)])."""
    from hyrise_amd import abi
    from hyrise_amd.operators import make_predicate
    nullable = make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_INT, -400, 300, nullable=True)
    floats = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, 80.0)
    product = (abi.ARITH_MUL, "cf", (abi.ARITH_SUB, (abi.TYPE_INT, 1), "gf"))
    full = [(abi.AGG_SUM, product), (abi.AGG_AVG, product), (abi.AGG_SUM, (abi.ARITH_ADD, "ci", (abi.TYPE_LONG, 7))), (abi.AGG_MIN, "ci"), (abi.AGG_MAX, "cf"),
            (abi.AGG_COUNT, "ci"), (abi.AGG_COUNT, None)]
    nothing = make_predicate(abi.PRED_GREATER_THAN, abi.TYPE_FLOAT, 1000.0)
    return {"fused_two_keys_general": ([("ci", nullable), ("cf", floats)], ["g1", "g2"], full),
            "fused_small_domain_slots": ([("cf", floats)], ["g3", "g2"], full),
            "fused_immediate_key": ([("ci", nullable)], ["g3"], full[:4]),
            "fused_no_groupby": ([("ci", nullable), ("cf", floats)], [], full),
            "fused_nothing_passes": ([("cf", nothing)], [], full[:1] + full[-1:]),
            "fused_nothing_passes_grouped": ([("cf", nothing)], ["g3"], full[-1:])}


def bind_tree(tree, column_of):
    if isinstance(tree, str):
        return column_of(tree)
    if tree is None or len(tree) == 2:
        return tree
    return (tree[0], bind_tree(tree[1], column_of), bind_tree(tree[2], column_of))


def pairs_of(left_pos, right_pos):
    left = left_pos.cpu().numpy().astype(np.uint32)
    right = right_pos.cpu().numpy().astype(np.uint32) if right_pos is not None else None
    out = []
    for i in range(left.shape[0]):
        l = None if left[i, 1] == 0xFFFFFFFF else (int(left[i, 0]), int(left[i, 1]))
        r = None
        if right is not None:
            r = None if right[i, 1] == 0xFFFFFFFF else (int(right[i, 0]), int(right[i, 1]))
        out.append((l, r))
    return out


def worker(rank, world, init_file, out_dir, executor_kind):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from hyrise_amd import abi
    from hyrise_amd.distributed import Comm, shard_column, sharded_aggregate, sharded_join_broadcast, sharded_join_repartition
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
    d = make_data()
    out = {"aggregate": {}, "joins": {}}
    shards = {}

    def shard(name):
        if name is None:
            return None
        if name not in shards:
            host, first = shard_column(d[name], world, rank)
            shards[name] = (ex.column(host), first)
        return shards[name][0]

    for case, (keys, aggregates) in aggregate_specs(d).items():
        first_chunk = shard_column(d[keys[0] if keys else "ci"], world, rank)[1]
        out["aggregate"][case] = sharded_aggregate(comm, ex, [shard(k) for k in keys], [(f, shard(c)) for f, c in aggregates], first_chunk)
    from hyrise_amd.distributed import sharded_scan_project_aggregate
    first_chunk = shard_column(d["ci"], world, rank)[1]
    for case, (filters, keys, aggregates) in fused_specs(d).items():
        out["aggregate"][case] = sharded_scan_project_aggregate(comm, ex, [(shard(c), p) for c, p in filters], [shard(k) for k in keys],
                                                                [(f, bind_tree(tree, shard)) for f, tree in aggregates], first_chunk)
    first_probe = shard_column(d["probe"], world, rank)[1]
    first_build = shard_column(d["build"], world, rank)[1]
    for mode in (abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI):
        if mode == abi.JOIN_INNER:   # build = left (smaller)
            b, p = sharded_join_broadcast(comm, ex, shard("build"), shard("probe"), mode, first_probe, d["chunk"], build_is_left=True)
            out["joins"][("broadcast", mode)] = pairs_of(b, p)
        else:                        # Left / Semi: the build side is the RIGHT input
            b, p = sharded_join_broadcast(comm, ex, shard("build"), shard("probe"), mode, first_probe, d["chunk"], build_is_left=False)
            out["joins"][("broadcast", mode)] = pairs_of(p, b)   # (left = probe, right = build)
    l, r = sharded_join_repartition(comm, ex, shard("build"), shard("probe"), first_build, first_probe, abi.JOIN_INNER)
    out["joins"][("repartition", abi.JOIN_INNER)] = pairs_of(l, r)
    first_dup = shard_column(d["dup_build"], world, rank)[1]
    l, r = sharded_join_repartition(comm, ex, shard("dup_build"), shard("probe"), first_dup, first_probe, abi.JOIN_INNER)
    out["joins"][("repartition_duplicates", abi.JOIN_INNER)] = pairs_of(l, r)
    l, r = sharded_join_repartition(comm, ex, shard("probe"), shard("dup_build"), first_probe, first_dup, abi.JOIN_SEMI)
    out["joins"][("repartition_semi", abi.JOIN_SEMI)] = pairs_of(l, r)
    # the modes that keep rows with NULL keys: NULL-key rows never travel, their own rank emits them
    first_nullable = shard_column(d["nullable_probe"], world, rank)[1]
    for mode in (abi.JOIN_LEFT, abi.JOIN_ANTI_NULL_AS_FALSE, abi.JOIN_ANTI_NULL_AS_TRUE):
        l, r = sharded_join_repartition(comm, ex, shard("nullable_probe"), shard("build"), first_nullable, first_build, mode)
        out["joins"][("repartition_outer", mode)] = pairs_of(l, r)
    l, r = sharded_join_repartition(comm, ex, shard("build"), shard("nullable_probe"), first_build, first_nullable, abi.JOIN_RIGHT)
    out["joins"][("repartition_outer", abi.JOIN_RIGHT)] = pairs_of(l, r)
    l, r = sharded_join_repartition(comm, ex, shard("nullable_probe"), shard("dup_build"), first_nullable, first_dup, abi.JOIN_ANTI_NULL_AS_TRUE)
    out["joins"][("repartition_anti_null_build", abi.JOIN_ANTI_NULL_AS_TRUE)] = pairs_of(l, r)
    first_even = shard_column(d["even_build"], world, rank)[1]
    for mode in (abi.JOIN_LEFT, abi.JOIN_ANTI_NULL_AS_FALSE, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_SEMI):
        l, r = sharded_join_repartition(comm, ex, shard("probe"), shard("even_build"), first_probe, first_even, mode)
        out["joins"][("repartition_empty_side", mode)] = pairs_of(l, r)
    l, r = sharded_join_repartition(comm, ex, shard("even_build"), shard("probe"), first_even, first_probe, abi.JOIN_RIGHT)
    out["joins"][("repartition_empty_side", abi.JOIN_RIGHT)] = pairs_of(l, r)
    try:   # a broadcast of the side hy_join_hash would PROBE with is refused (every rank would emit the gathered side's rows)
        sharded_join_broadcast(comm, ex, shard("build"), shard("probe"), abi.JOIN_LEFT, first_probe, d["chunk"], build_is_left=True)
        out["broadcast_refused"] = False
    except NotImplementedError:
        out["broadcast_refused"] = True
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def check_results(results):
    """Against the single-process oracle: aggregates in the reference's order on every rank; joins as multisets over the ranks."""
    from hyrise_amd import abi
    from support import column_values, join_result_multiset, oracle_aggregate, oracle_join
    d = make_data()
    for case, (keys, aggregates) in aggregate_specs(d).items():
        groupby = [d[k] for k in keys]
        want = oracle_aggregate(groupby, [(f, d[c] if c else None) for f, c in aggregates])
        key_values = [column_values(g) for g in groupby]
        flat, offset = {}, 0
        shape = groupby[0] if groupby else d["ci"]
        for c, seg in enumerate(shape.segments):
            for i in range(seg.size):
                flat[(c, i)] = offset + i
            offset += seg.size
        for rank_rows in (r["aggregate"][case] for r in results):   # every rank ends with the same merged result
            assert len(rank_rows) == want.n_groups, case
            for g, (key, cells) in enumerate(rank_rows):
                rid = tuple(int(x) for x in want.row_ids[g])
                expected_key = tuple(kv[flat[rid]] for kv in key_values)
                assert tuple(key) == expected_key, f"{case}: group {g} is {key}, the reference has {expected_key} there (group order)"
                for a in range(len(aggregates)):
                    expected, got = want.column(a)[g], cells[a]
                    if expected is None:
                        assert got is None, (case, a)
                    elif isinstance(expected, float):
                        assert abs(got - expected) <= 1e-9 * max(1.0, abs(expected)), (case, a, got, expected)
                    else:
                        assert got == expected, (case, a, got, expected)

    # the sharded fused pass against the single-process operator chain on the oracle
    from support import oracle_chain
    for case, (filters, keys, aggregates) in fused_specs(d).items():
        want, base_rows, sizes = oracle_chain([(d[c], p) for c, p in filters], [d[k] for k in keys], [(f, bind_tree(tree, lambda name: d[name])) for f, tree in aggregates])
        key_values = [column_values(d[k]) for k in keys]
        first_of_chunk = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        for rank_rows in (r["aggregate"][case] for r in results):
            assert len(rank_rows) == want.n_groups, case
            for g, (key, cells) in enumerate(rank_rows):
                if keys:
                    in_chain = want.row_ids[g]
                    base = base_rows[int(first_of_chunk[int(in_chain[0])]) + int(in_chain[1])]
                    flat_row = int(base[0]) * d["chunk"] + int(base[1])
                    expected_key = tuple(kv[flat_row] for kv in key_values)
                    assert tuple(key) == expected_key, f"{case}: group {g} is {key}, the chain has {expected_key} there (group order)"
                for a in range(len(aggregates)):
                    expected, got = want.column(a)[g], cells[a]
                    if expected is None:
                        assert got is None, (case, a, got)
                    elif isinstance(expected, float):
                        assert abs(got - expected) <= 1e-9 * max(1.0, abs(expected)), (case, a, got, expected)
                    else:
                        assert got == expected, (case, a, got, expected)

    def multiset(name, mode):
        got = []
        for r in results:
            got += r["joins"][(name, mode)]
        return sorted(got, key=lambda p: (p[0] is None, p[0] or (0, 0), p[1] is None, p[1] or (0, 0)))

    inner = oracle_join(d["build"], d["probe"], abi.JOIN_INNER)
    assert multiset("broadcast", abi.JOIN_INNER) == join_result_multiset(inner, abi.JOIN_INNER)
    assert multiset("repartition", abi.JOIN_INNER) == join_result_multiset(inner, abi.JOIN_INNER)
    for mode in (abi.JOIN_LEFT, abi.JOIN_SEMI):
        whole = oracle_join(d["probe"], d["build"], mode)
        assert multiset("broadcast", mode) == join_result_multiset(whole, mode), f"mode {mode}"
    dup = oracle_join(d["dup_build"], d["probe"], abi.JOIN_INNER)
    assert multiset("repartition_duplicates", abi.JOIN_INNER) == join_result_multiset(dup, abi.JOIN_INNER)
    semi = oracle_join(d["probe"], d["dup_build"], abi.JOIN_SEMI)
    assert multiset("repartition_semi", abi.JOIN_SEMI) == join_result_multiset(semi, abi.JOIN_SEMI)
    for mode in (abi.JOIN_LEFT, abi.JOIN_ANTI_NULL_AS_FALSE, abi.JOIN_ANTI_NULL_AS_TRUE):
        whole = oracle_join(d["nullable_probe"], d["build"], mode)
        assert multiset("repartition_outer", mode) == join_result_multiset(whole, mode), f"repartitioned join, mode {mode}"
    whole = oracle_join(d["build"], d["nullable_probe"], abi.JOIN_RIGHT)
    assert multiset("repartition_outer", abi.JOIN_RIGHT) == join_result_multiset(whole, abi.JOIN_RIGHT)
    whole = oracle_join(d["nullable_probe"], d["dup_build"], abi.JOIN_ANTI_NULL_AS_TRUE)   # (a NULL key on the right side: nothing qualifies)
    assert whole.n_pairs == 0 and multiset("repartition_anti_null_build", abi.JOIN_ANTI_NULL_AS_TRUE) == []
    for mode in (abi.JOIN_LEFT, abi.JOIN_ANTI_NULL_AS_FALSE, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_SEMI):
        whole = oracle_join(d["probe"], d["even_build"], mode)
        assert multiset("repartition_empty_side", mode) == join_result_multiset(whole, mode), f"a rank without build tuples, mode {mode}"
    whole = oracle_join(d["even_build"], d["probe"], abi.JOIN_RIGHT)
    assert multiset("repartition_empty_side", abi.JOIN_RIGHT) == join_result_multiset(whole, abi.JOIN_RIGHT)
    assert all(r["broadcast_refused"] for r in results)
