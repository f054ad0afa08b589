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
    assert star_was_fused() == 2   # (one pass over lineorder, the survivors grouped inside it: csrc/join_star.hpp star_finish) ...
    from hyrise_amd import abi
    finished = result_bytes(result, len(star_aggregates))
    with abi.option(abi.OPT_STAR_FUSED_FINISH, 0):   # ... the survivors' RowIDs written and hy_aggregate_hash over the exported columns: the same bytes
        result, star_joined = star_join_aggregate(dimensions, star_groupby, star_aggregates)
    assert star_was_fused() == 1 and star_joined == joined and result_bytes(result, len(star_aggregates)) == finished
    with abi.option(abi.OPT_STAR_FUSED_PROBE, 0):    # ... and dimension by dimension (the groups in the order of the last join's rows)
        result, star_joined = star_join_aggregate(dimensions, star_groupby, star_aggregates)
    assert star_was_fused() == 0 and star_joined == joined and ssb.result_rows(ssb.star_groups(result, len(star_groupby))) == sqlite


@pytest.mark.timeout(600)
def test_ssb_scale_factor_one_on_device(device):
    """Config 5 at scale factor 1 (6 000 000 lineorder rows, 92 chunks; part 200 000, customer 30 000, supplier 2 000 rows): the operator
    chain and the one-call plan on the device against the same plans on the CPU oracle (every thread) and against SQLite -- group rows,
    sums and joined-row counts are integers: equal, not close.  (SF30 itself is checked by bench.py against the oracle on every run.)"""
    import os
    from hyrise_amd import abi
    from hyrise_amd.operators import star_join_aggregate
    data = ssb.SsbData(scale_factor=1.0, seed=11)
    assert data.n_lineorder == 6_000_000
    host = data.host_columns()
    ex = HipExecutor(torch.device("cuda", 0))
    columns = {name: ex.column(c) for name, c in host.items()}
    oracle = OracleExecutor(threads=os.cpu_count() or 1)
    for query, sql in (("2.1", ssb.Q2_1_SQL), ("4.1", ssb.Q4_1_SQL)):
        groupby, aggregates, joined = ssb.run_query(ex, columns, query)
        got = ssb.result_rows(aggregate_groups(ex, groupby, aggregates))
        o_groupby, o_aggregates, o_joined = ssb.run_query(oracle, host, query)
        want = ssb.result_rows(aggregate_groups(oracle, o_groupby, o_aggregates))
        assert joined == o_joined and got == want, f"Q{query}: operator chain on the device vs the CPU oracle"
        dimensions, star_groupby, star_aggregates = ssb.star_plan(columns, query)
        same = {}
        for fused in (2, 1, 0):
            with abi.option(abi.OPT_STAR_FUSED_PROBE, 1 if fused else 0), abi.option(abi.OPT_STAR_FUSED_FINISH, 1 if fused == 2 else 0):
                result, star_joined = star_join_aggregate(dimensions, star_groupby, star_aggregates)
            assert star_was_fused() == fused
            assert star_joined == o_joined and ssb.result_rows(ssb.star_groups(result, len(star_groupby))) == want, f"Q{query}: hy_star_join_aggregate (fused {fused}) vs the CPU oracle"
            same[fused] = result_bytes(result, len(star_aggregates))
        assert same[2] == same[1], f"Q{query}: grouped inside the join vs hy_aggregate_hash over the exported survivors: group order, representative rows, cells"
        rows = data.sqlite_result(sql)
        sqlite = sorted(((r[1], r[2]), r[0]) for r in rows) if query == "2.1" else sorted(((r[0], r[1]), r[2]) for r in rows)
        assert got == sqlite, f"Q{query}: SQLite"


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


def star_was_fused():
    """0: one hy_join_hash per dimension; 1: every dimension probed in one pass; 2: ... and the survivors grouped inside it"""
    from hyrise_amd import abi
    lib = abi.load_library()
    lib.hy_debug_star_fused.restype = int
    return lib.hy_debug_star_fused()


def result_bytes(result, n_aggregates):
    """Everything hy_star_join_aggregate wrote: the groups in their order, representative RowIDs, every cell with its type and NULL flag."""
    n = result.n_groups
    return (n, result.row_ids[:n].tobytes(), [(result.columns[a].data_type, result.raw[a][:n].tobytes(), result.nulls[a][:n].tobytes()) for a in range(n_aggregates)])


@pytest.mark.parametrize("fused", [2, 1, 0], ids=["fused_finish", "fused_probe", "join_by_join"])
@pytest.mark.parametrize("case", ["filtered", "nothing_survives", "first_dimension_unfiltered", "dangling_foreign_keys"])
def test_star_join_aggregate_small_tables(device, options, case, fused):
    """hy_star_join_aggregate on tables small enough to join with numpy: two dimensions (either may be unfiltered), foreign keys without a
    partner, a filter nothing passes (an empty join result: no groups), GROUP BY a column of each dimension, SUM of a fact column,
    SUM of an expression over two fact columns, COUNT(*)."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import make_predicate, star_join_aggregate
    from hyrise_amd.storage import DeviceColumn
    rng = np.random.default_rng(len(case))
    n_fact, n_a, n_b = 200_000, 3_000, 500
    a_key = np.arange(1, n_a + 1, dtype=np.int32) * 3                       # sparse unique keys
    a_group = rng.integers(0, 7, n_a).astype(np.int32)
    a_filter = rng.integers(0, 10, n_a).astype(np.int32)
    b_key = rng.permutation(n_b).astype(np.int32) + 100                     # unsorted unique keys
    b_group = rng.integers(0, 5, n_b).astype(np.int32)
    b_filter = rng.integers(0, 4, n_b).astype(np.int32)
    fk_a = a_key[rng.integers(0, n_a, n_fact)].copy()
    fk_b = b_key[rng.integers(0, n_b, n_fact)].copy()
    if case == "dangling_foreign_keys":
        fk_a[rng.random(n_fact) < 0.3] = 1                                   # no such key
        fk_b[rng.random(n_fact) < 0.2] = 99
    x = rng.integers(-1000, 1000, n_fact).astype(np.int32)
    y = rng.integers(0, 50, n_fact).astype(np.int32)
    column = lambda values, encoding=abi.ENC_UNENCODED, chunk=20_000: DeviceColumn(storage.make_column(values, None, encoding, chunk))
    c = {"a_key": column(a_key, chunk=1_000), "a_group": column(a_group, abi.ENC_DICTIONARY, 1_000), "a_filter": column(a_filter, abi.ENC_FRAME_OF_REFERENCE, 1_000),
         "b_key": column(b_key, chunk=200), "b_group": column(b_group, abi.ENC_FRAME_OF_REFERENCE, 200), "b_filter": column(b_filter, abi.ENC_DICTIONARY, 200),
         "fk_a": column(fk_a, abi.ENC_FRAME_OF_REFERENCE), "fk_b": column(fk_b, abi.ENC_FRAME_OF_REFERENCE), "x": column(x, abi.ENC_FRAME_OF_REFERENCE), "y": column(y, abi.ENC_DICTIONARY)}
    a_value = 99 if case == "nothing_survives" else 3
    a_predicate = None if case == "first_dimension_unfiltered" else make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, a_value) if case != "nothing_survives" else make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, 99)
    dimensions = [(c["a_key"], None if a_predicate is None else c["a_filter"], a_predicate, c["fk_a"]),
                  (c["b_key"], c["b_filter"], make_predicate(abi.PRED_NOT_EQUALS, abi.TYPE_INT, 2), c["fk_b"])]
    groupby = [(1, c["a_group"]), (2, c["b_group"])]
    aggregates = [(abi.AGG_SUM, (0, c["x"]), None, None), (abi.AGG_SUM, (0, c["x"]), abi.ARITH_MUL, (0, c["y"])), (abi.AGG_COUNT, None, None, None),
                  (abi.AGG_MIN, groupby[0], None, None), (abi.AGG_MIN, groupby[1], None, None)]
    options.set(abi.OPT_STAR_FUSED_PROBE, 1 if fused else 0)
    options.set(abi.OPT_STAR_FUSED_FINISH, 1 if fused == 2 else 0)
    result, joined = star_join_aggregate(dimensions, groupby, aggregates)
    assert star_was_fused() == fused   # (these tables are the fused probe's shape: int32 keys, unique per dimension, FrameOfReference foreign keys)
    if fused == 2:   # the same bytes as hy_aggregate_hash over the exported survivors: group order, representative rows, types, cells
        options.set(abi.OPT_STAR_FUSED_FINISH, 0)
        other, _ = star_join_aggregate(dimensions, groupby, aggregates)
        assert star_was_fused() == 1 and result_bytes(other, len(aggregates)) == result_bytes(result, len(aggregates))
    # numpy: positions of the partners, then the filters
    a_of = {int(k): i for i, k in enumerate(a_key)}
    b_of = {int(k): i for i, k in enumerate(b_key)}
    ia = np.array([a_of.get(int(k), -1) for k in fk_a])
    ib = np.array([b_of.get(int(k), -1) for k in fk_b])
    keep = (ia >= 0) & (ib >= 0)
    if case == "nothing_survives":
        keep &= False
    elif case != "first_dimension_unfiltered":
        keep &= a_filter[np.maximum(ia, 0)] < 3
    keep &= b_filter[np.maximum(ib, 0)] != 2
    assert joined == int(keep.sum())
    want = {}
    for ga, gb, xv, yv in zip(a_group[ia[keep]], b_group[ib[keep]], x[keep].astype(np.int64), y[keep].astype(np.int64)):
        cell = want.setdefault((int(ga), int(gb)), [0, 0, 0])
        cell[0] += int(xv)
        cell[1] += int(xv) * int(yv)
        cell[2] += 1
    n = result.n_groups
    got = {(result.column(3)[i], result.column(4)[i]): [result.column(0)[i], result.column(1)[i], result.column(2)[i]] for i in range(n)}
    assert got == want and (case != "nothing_survives" or n == 0)


def test_star_join_aggregate_with_null_cells(device):
    """NULLs in what the plan reads (round 6: the plan's intermediate tables carry null vectors; before, such a plan was refused): a foreign key
    that is NULL finds no partner -- it is not joined as key 0 --, a NULL aggregate input is skipped, a NULL GROUP BY cell is a group of its
    own.  The fused probe does not read nullable foreign keys and star_finish carries no NULLs: these plans run join by join / on the RowID
    path by themselves.  Against numpy."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import star_join_aggregate
    from hyrise_amd.storage import DeviceColumn
    rng = np.random.default_rng(3)
    n = 50_000
    a_key = np.arange(0, 200, dtype=np.int32)
    a_group = rng.integers(0, 6, 200).astype(np.int32)
    a_group_nulls = rng.random(200) < 0.1
    b_key = np.arange(0, 50, dtype=np.int32)                                  # (key 0 exists: a NULL exported as 0 would find it)
    fk_a = rng.integers(0, 200, n).astype(np.int32)
    fk_b = rng.integers(0, 50, n).astype(np.int32)
    x = rng.integers(0, 100, n).astype(np.int32)
    nulls = rng.random(n) < 0.1
    column = lambda values, null=None, chunk=8_000: DeviceColumn(storage.make_column(values, null, abi.ENC_UNENCODED, chunk))
    a, b, fa, x_plain = column(a_key, None, 64), column(b_key, None, 64), column(fk_a), column(x)
    group_plain, group_nullable = column(a_group, None, 64), column(a_group, a_group_nulls, 64)
    dimensions = lambda second_key: [(a, None, None, fa), (b, None, None, second_key)]

    def run(second_key, measure, group):
        groupby = [(1, group)]
        result, joined = star_join_aggregate(dimensions(second_key), groupby, [(abi.AGG_SUM, (0, measure), None, None), (abi.AGG_COUNT, (0, measure), None, None), (abi.AGG_COUNT, None, None, None),
                                                                                (abi.AGG_MIN, groupby[0], None, None)])
        return {result.column(3)[i]: (result.column(0)[i], result.column(1)[i], result.column(2)[i]) for i in range(result.n_groups)}, joined

    def expected(key_nulls, measure_nulls, group_nulls):
        keep = ~key_nulls if key_nulls is not None else np.ones(n, dtype=bool)
        want = {}
        for row in np.flatnonzero(keep):
            g = None if group_nulls is not None and group_nulls[fk_a[row]] else int(a_group[fk_a[row]])
            cell = want.setdefault(g, [None, 0, 0])
            cell[2] += 1
            if measure_nulls is None or not measure_nulls[row]:
                cell[0] = (cell[0] or 0) + int(x[row])
                cell[1] += 1
        return {g: tuple(c) for g, c in want.items()}, int(keep.sum())

    assert run(column(fk_b, nulls), x_plain, group_plain) == expected(nulls, None, None) and star_was_fused() == 0   # a nullable foreign key: join by join
    assert run(column(fk_b), column(x, nulls), group_plain) == expected(None, nulls, None) and star_was_fused() == 1  # NULL inputs: star_finish hands over to the RowID path
    assert run(column(fk_b), x_plain, group_nullable) == expected(None, None, a_group_nulls) and star_was_fused() == 1   # a NULL group
    assert run(column(fk_b, nulls), column(x, nulls), group_nullable) == expected(nulls, nulls, a_group_nulls)
    got, joined = run(column(fk_b), x_plain, group_plain)
    assert joined == n and len(got) == 6 and star_was_fused() == 2


def test_star_join_shapes_the_fused_probe_leaves_to_the_joins(device):
    """A dimension whose key comes twice (not a primary key: every partner counts, JoinHash's business), a dimension with a key range of
    2^27 values (no direct table), and more dimensions than fit LDS at once with one of them asked in global memory: the first two run
    join by join, the last fused -- all against numpy."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import make_predicate, star_join_aggregate
    from hyrise_amd.storage import DeviceColumn
    rng = np.random.default_rng(5)
    n_fact = 150_000
    column = lambda values, encoding=abi.ENC_UNENCODED, chunk=20_000: DeviceColumn(storage.make_column(np.ascontiguousarray(values), None, encoding, chunk))

    def run(dim_keys, dim_groups, foreign):
        dims = [(column(k, chunk=5_000), None, None, column(f, abi.ENC_FRAME_OF_REFERENCE)) for k, f in zip(dim_keys, foreign)]
        group_columns = [column(g, abi.ENC_FRAME_OF_REFERENCE, 5_000) for g in dim_groups]
        groupby = [(d + 1, g) for d, g in enumerate(group_columns)]
        result, joined = star_join_aggregate(dims, groupby, [(abi.AGG_COUNT, None, None, None)] + [(abi.AGG_MIN, g, None, None) for g in groupby])
        got = {tuple(result.column(1 + d)[i] for d in range(len(dims))): result.column(0)[i] for i in range(result.n_groups)}
        return got, joined

    def expected(dim_keys, dim_groups, foreign):
        partners = []
        for keys, groups, fk in zip(dim_keys, dim_groups, foreign):
            of = {}
            for k, g in zip(keys.tolist(), groups.tolist()):
                of.setdefault(k, []).append(g)
            partners.append([of.get(k, []) for k in fk.tolist()])
        want, joined = {}, 0
        for row in range(n_fact):
            combos = [()]
            for p in partners:
                combos = [c + (g,) for c in combos for g in p[row]]
            for c in combos:
                want[c] = want.get(c, 0) + 1
                joined += 1
        return want, joined

    # a key twice
    keys = [np.concatenate([np.arange(1, 400, dtype=np.int32), np.array([7, 7, 9], dtype=np.int32)]), np.arange(1, 50, dtype=np.int32)]
    groups = [rng.integers(0, 4, len(keys[0])).astype(np.int32), rng.integers(0, 3, len(keys[1])).astype(np.int32)]
    foreign = [rng.integers(1, 420, n_fact).astype(np.int32), rng.integers(1, 55, n_fact).astype(np.int32)]
    assert run(keys, groups, foreign) == expected(keys, groups, foreign) and star_was_fused() == 0
    # a sparse key range
    keys = [np.array(sorted(rng.choice(1 << 27, 300, replace=False)), dtype=np.int32), np.arange(1, 50, dtype=np.int32)]
    groups = [rng.integers(0, 4, 300).astype(np.int32), rng.integers(0, 3, 49).astype(np.int32)]
    foreign = [keys[0][rng.integers(0, 300, n_fact)], rng.integers(1, 55, n_fact).astype(np.int32)]
    assert run(keys, groups, foreign) == expected(keys, groups, foreign) and star_was_fused() == 0
    # five dimensions, one of them with 1.5 M key values (more than the others leave of LDS): fused, that one asked in global memory
    sizes = [1_500_000, 1_000, 300, 50, 7]
    keys = [np.arange(10, 10 + n, dtype=np.int32) for n in sizes]
    keep = rng.random(sizes[0]) < 0.5
    keys[0] = keys[0][keep]                      # (half of the big dimension's keys exist)
    groups = [(k % 3).astype(np.int32) for k in keys]
    foreign = [rng.integers(5, 15 + n, n_fact).astype(np.int32) for n in sizes]
    assert run(keys, groups, foreign) == expected(keys, groups, foreign) and star_was_fused() == 1   # (five GROUP BY columns: not star_finish's shape)


@pytest.mark.parametrize("layout", ["int32", "for32", "for16", "for8", "two_in_memory"])
def test_star_join_streams_the_first_dimension_that_does_not_fit_lds(device, layout):
    """Three dimensions, one with 1.5 M key values (its presence bits do not fit beside the others'): the probe streams its foreign keys with the
    LDS-resident dimensions' and asks the surviving rows' bits in global memory -- per width of the stored foreign keys (plain int32; frame of
    reference with 4-, 2- and 1-byte offsets: keys clustered per block narrow the offsets) and with a second such dimension behind it (that one
    asked row by row).  A ragged last tile; against numpy."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import star_join_aggregate
    from hyrise_amd.storage import DeviceColumn
    rng = np.random.default_rng(11)
    n_fact = 150_001
    column = lambda values, encoding=abi.ENC_UNENCODED, chunk=20_000: DeviceColumn(storage.make_column(np.ascontiguousarray(values), None, encoding, chunk))
    sizes = [1_500_000, 1_500_000, 300] if layout == "two_in_memory" else [1_500_000, 1_000, 300]
    keys = [np.arange(10, 10 + n, dtype=np.int32) for n in sizes]
    for d, n in enumerate(sizes):
        if n > 100_000:
            keys[d] = keys[d][rng.random(n) < 0.5]     # (half of a big dimension's keys exist)
    groups = [(k % 3).astype(np.int32) for k in keys]
    foreign = [rng.integers(5, 15 + n, n_fact).astype(np.int32) for n in sizes]
    if layout in ("for16", "for8"):                    # (ascending with small steps: a block of 2048 rows spans < 65536 / < 256 values)
        step = 9 if layout == "for16" else 0.1
        foreign[0] = (5 + np.floor(np.arange(n_fact) * step) + rng.integers(0, 20, n_fact)).astype(np.int32)
    encoding = abi.ENC_UNENCODED if layout == "int32" else abi.ENC_FRAME_OF_REFERENCE
    dims = [(column(k, chunk=50_000), None, None, column(f, encoding)) for k, f in zip(keys, foreign)]
    if layout in ("for16", "for8"):
        stored = storage.make_column(np.ascontiguousarray(foreign[0]), None, abi.ENC_FRAME_OF_REFERENCE, 20_000)
        widths = {segment.width for segment in stored.segments}
        assert widths == {2 if layout == "for16" else 1}, widths
    group_columns = [column(g, abi.ENC_FRAME_OF_REFERENCE, 50_000) for g in groups]
    groupby = [(d + 1, g) for d, g in enumerate(group_columns)]
    result, joined = star_join_aggregate(dims, groupby, [(abi.AGG_COUNT, None, None, None)] + [(abi.AGG_MIN, g, None, None) for g in groupby])
    assert star_was_fused() == 2
    got = {tuple(int(result.column(1 + d)[i]) for d in range(3)): int(result.column(0)[i]) for i in range(result.n_groups)}
    alive = np.ones(n_fact, dtype=bool)
    codes = []
    for k, fk in zip(keys, foreign):
        position = np.searchsorted(k, fk)
        found = (position < len(k)) & (k[np.minimum(position, len(k) - 1)] == fk)
        alive &= found
        codes.append(fk % 3)
    want = {}
    for c in zip(*(code[alive].tolist() for code in codes)):
        want[c] = want.get(c, 0) + 1
    assert joined == int(alive.sum()) and joined > 1000
    assert got == want


@pytest.mark.parametrize("case", ["immediate_key", "three_thousand_groups", "six_thousand_groups", "long_columns", "division_by_zero"])
def test_star_finish_shapes(device, options, case):
    """star_finish (the aggregate inside the fused probe) against hy_aggregate_hash over the exported survivors -- the same bytes -- and numpy:
    one int GROUP BY column with a dense key range (the immediate-key shortcut: ascending keys, the LAST row represents a group), more
    groups than a workgroup's LDS table holds (rows go to the global table), more than the compacted result holds (refused: the RowID
    path answers), int64 columns, and an expression whose cell is NULL (x / 0: refused, the RowID path answers)."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import make_predicate, star_join_aggregate
    from hyrise_amd.storage import DeviceColumn
    rng = np.random.default_rng(len(case))
    n_fact, n_a, n_b = 400_000, 8_000, 40
    column = lambda values, encoding=abi.ENC_UNENCODED, chunk=30_000: DeviceColumn(storage.make_column(np.ascontiguousarray(values), None, encoding, chunk))
    n_groups = {"immediate_key": 25, "three_thousand_groups": 1_000, "six_thousand_groups": 2_000}.get(case, 300)
    wide = np.int64 if case == "long_columns" else np.int32
    a_key = rng.permutation(n_a).astype(np.int32) + 17
    a_group = (rng.integers(0, n_groups, n_a).astype(wide) - 3) * (1 << 33 if case == "long_columns" else 1)
    a_filter = rng.integers(0, 10, n_a).astype(np.int32)
    b_key = np.arange(n_b, dtype=np.int32) * 2
    b_group = rng.integers(0, 3, n_b).astype(np.int32)
    fk_a = a_key[rng.integers(0, n_a, n_fact)]
    fk_b = rng.integers(0, 2 * n_b, n_fact).astype(np.int32)              # odd keys have no partner
    x = (rng.integers(-50_000, 50_000, n_fact).astype(wide)) * (1 << 20 if case == "long_columns" else 1)
    y = rng.integers(0 if case == "division_by_zero" else 1, 60, n_fact).astype(np.int32)
    c = {"a_key": column(a_key, chunk=1_000), "a_group": column(a_group, abi.ENC_DICTIONARY if wide is np.int32 else abi.ENC_UNENCODED, 1_000), "a_filter": column(a_filter, abi.ENC_DICTIONARY, 1_000),
         "b_key": column(b_key, chunk=16), "b_group": column(b_group, abi.ENC_FRAME_OF_REFERENCE, 16), "fk_a": column(fk_a, abi.ENC_FRAME_OF_REFERENCE), "fk_b": column(fk_b),
         "x": column(x, abi.ENC_UNENCODED if wide is np.int64 else abi.ENC_FRAME_OF_REFERENCE), "y": column(y, abi.ENC_DICTIONARY)}
    dimensions = [(c["a_key"], c["a_filter"], make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 8), c["fk_a"]), (c["b_key"], None, None, c["fk_b"])]
    groupby = [(1, c["a_group"])] if case in ("immediate_key", "long_columns") else [(1, c["a_group"]), (2, c["b_group"])]
    op = abi.ARITH_DIV if case == "division_by_zero" else abi.ARITH_SUB
    aggregates = [(abi.AGG_SUM, (0, c["x"]), op, (0, c["y"])), (abi.AGG_MIN, (0, c["x"]), None, None), (abi.AGG_MAX, (0, c["y"]), None, None), (abi.AGG_AVG, (0, c["x"]), None, None),
                  (abi.AGG_COUNT, (0, c["y"]), None, None), (abi.AGG_COUNT, None, None, None), (abi.AGG_MIN, (1, c["a_group"]), None, None), (abi.AGG_MAX, (2, c["b_group"]), None, None)]
    capacity = 32_768
    result, joined = star_join_aggregate(dimensions, groupby, aggregates, group_capacity=capacity)
    assert star_was_fused() == (1 if case in ("six_thousand_groups", "division_by_zero") else 2)   # (refused: the RowID path answered)
    options.set(abi.OPT_STAR_FUSED_FINISH, 0)
    other, other_joined = star_join_aggregate(dimensions, groupby, aggregates, group_capacity=capacity)
    assert star_was_fused() == 1 and other_joined == joined and result_bytes(other, len(aggregates)) == result_bytes(result, len(aggregates))
    if case == "division_by_zero":   # (x / 0 is a NULL cell the chain's SUM skips; star_finish carries no NULLs and leaves the plan to the chain)
        assert result.n_groups > 0
        return
    # numpy
    ia = np.full(int(a_key.max()) + 1, -1, dtype=np.int64)
    ia[a_key] = np.arange(n_a)
    rows_a = ia[fk_a]
    keep = (a_filter[rows_a] < 8) & (fk_b % 2 == 0)
    assert joined == int(keep.sum())
    ga, gb = a_group[rows_a][keep].astype(np.int64), b_group[fk_b[keep] // 2].astype(np.int64)
    key = ga if len(groupby) == 1 else ga * 8 + gb
    xs, ys = x[keep].astype(np.int64), y[keep].astype(np.int64)
    diff = (x[keep] - y[keep].astype(wide)).astype(np.int64)               # (int - int is computed in int32: these values do not wrap)
    want = {}
    for k, d, xv, yv, gav, gbv in zip(key.tolist(), diff.tolist(), xs.tolist(), ys.tolist(), ga.tolist(), gb.tolist()):
        cell = want.setdefault(k, [0, xv, yv, 0, 0, 0, gav, gbv])
        cell[0] += d
        cell[1] = min(cell[1], xv)
        cell[2] = max(cell[2], yv)
        cell[3] += xv
        cell[4] += 1
        cell[5] += 1
        cell[7] = max(cell[7], gbv)
    n = result.n_groups
    assert n == len(want)
    columns = [result.column(a) for a in range(len(aggregates))]
    for i in range(n):
        k = columns[6][i] if len(groupby) == 1 else columns[6][i] * 8 + columns[7][i]
        w = want[k]
        assert [columns[a][i] for a in (0, 1, 2, 4, 5, 6)] == [w[0], w[1], w[2], w[4], w[5], w[6]] and columns[3][i] == w[3] / w[4]
        assert len(groupby) == 1 or columns[7][i] == w[7]
    if case == "immediate_key":
        assert columns[6] == sorted(columns[6])                            # ascending keys


@pytest.mark.parametrize("key_layout", ["bit_packed_dictionary", "run_length", "bit_packed_frame_of_reference"])
def test_star_join_with_a_compressed_dimension_key_takes_the_join_chain(device, key_layout):
    """A dimension key held as a RunLength segment or a BitPackingVector is not decoded by the fused probe's cell reader: such a plan must
    run join by join (hy_join_hash reads the key's decoded twin) and give numpy's groups -- not garbage keys."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.operators import star_join_aggregate
    from hyrise_amd.storage import DeviceColumn, HostColumn
    rng = np.random.default_rng(11)
    n_fact, n_a, n_b = 120_000, 2_000, 300
    a_key = np.arange(5, 5 + n_a, dtype=np.int32)
    a_group = rng.integers(0, 6, n_a).astype(np.int32)
    b_key = np.arange(1, n_b + 1, dtype=np.int32)
    fk_a = rng.integers(0, n_a + 40, n_fact).astype(np.int32)               # some foreign keys without a partner
    fk_b = rng.integers(1, n_b + 1, n_fact).astype(np.int32)
    x = rng.integers(0, 1000, n_fact).astype(np.int32)
    if key_layout == "run_length":
        host_key = HostColumn([storage.encode_run_length(a_key[i:i + 500], None) for i in range(0, n_a, 500)], abi.TYPE_INT)
    else:
        plain = storage.make_column(a_key, None, abi.ENC_DICTIONARY if key_layout == "bit_packed_dictionary" else abi.ENC_FRAME_OF_REFERENCE, 500)
        host_key = HostColumn([storage.bit_pack_segment(s) for s in plain.segments], abi.TYPE_INT)
    column = lambda values, encoding=abi.ENC_UNENCODED, chunk=20_000: DeviceColumn(storage.make_column(values, None, encoding, chunk))
    a, a_g, b = DeviceColumn(host_key), column(a_group, abi.ENC_DICTIONARY, 500), column(b_key, chunk=100)
    dimensions = [(a, None, None, column(fk_a, abi.ENC_FRAME_OF_REFERENCE)), (b, None, None, column(fk_b, abi.ENC_FRAME_OF_REFERENCE))]
    groupby = [(1, a_g)]
    result, joined = star_join_aggregate(dimensions, groupby, [(abi.AGG_SUM, (0, column(x)), None, None), (abi.AGG_COUNT, None, None, None), (abi.AGG_MIN, groupby[0], None, None)])
    assert star_was_fused() == 0
    ia = fk_a - 5
    keep = (ia >= 0) & (ia < n_a)
    assert joined == int(keep.sum())
    want = {}
    for g, xv in zip(a_group[ia[keep]], x[keep].astype(np.int64)):
        cell = want.setdefault(int(g), [0, 0])
        cell[0] += int(xv)
        cell[1] += 1
    got = {result.column(2)[i]: [result.column(0)[i], result.column(1)[i]] for i in range(result.n_groups)}
    assert got == want
