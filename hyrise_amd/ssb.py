"""Star Schema Benchmark (config 5 of BASELINE.json): synthetic tables per the SSB specification and the star-join queries
Q2.1 and Q4.1 as chains of the three hot-path operators.

The reference generates SSB with ssb-dbgen, which is not in its tree; the tables here follow the
specification's shapes (resources/benchmark/ssb/schema/*.csv.json for the columns): lineorder ~ 6 M x SF rows with uniform
foreign keys, customer 30 000 x SF, supplier 2 000 x SF, part 200 000 x floor(1 + log2 SF), date = the 2 557 days of
1992-1998 (the specification's 2 556 plus a leap day).  String attributes (p_category 'MFGR#12', s_region 'AMERICA', ...) are generated as the integer codes a
dictionary scan sees (the l_shipdate precedent of SURVEY.md section 8: the kernel compares value ids either way).

Queries (resources/benchmark/ssb/queries/2.1.sql, 4.1.sql), planned the way Hyrise's optimizer plans a star join -- scan
the dimensions, then one PK-FK JoinHash per dimension with the filtered dimension as build side, then AggregateHash:

  Q2.1  sum(lo_revenue) group by d_year, p_brand1   where p_category = 'MFGR#12' and s_region = 'AMERICA'
  Q4.1  sum(lo_revenue - lo_supplycost) group by d_year, c_nation
        where c_region = 'AMERICA' and s_region = 'AMERICA' and p_mfgr in ('MFGR#1', 'MFGR#2')

Everything between the dimension scans and the final groups stays on the device: join outputs are PosLists in HBM, the next
operator reads the base columns through them as ReferenceSegments (hy_column_create over device memory), and a join over
such a reference table is dereferenced with hy_gather_row_ids exactly like write_output_chunks does for reference inputs
(join_output_writing.cpp:95-200).  `run_query` is written against the executor interface of hyrise_amd/distributed.py, so
the same plan runs on the HIP library, on the CPU oracle (tests) and -- lineorder chunk-sharded, dimensions replicated or
hash-repartitioned -- on several GPUs.
"""
import math

import numpy as np

from . import abi, storage

REGIONS = ("AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST")
AMERICA = 1
VIRTUAL_CHUNK = 65535   # operator results are presented to the next operator as tables of this chunk size
DENSE_CHUNK = 65536     # ... and a materialised foreign-key column of the join result in chunks that start on 16-byte boundaries

Q2_1_SQL = """select sum(lo_revenue), d_year, p_brand1 from lineorder, "date", part, supplier
 where lo_orderdate = d_datekey and lo_partkey = p_partkey and lo_suppkey = s_suppkey and p_category = 12 and s_region = 1
 group by d_year, p_brand1 order by d_year, p_brand1"""
Q4_1_SQL = """select d_year, c_nation, sum(lo_revenue - lo_supplycost) as profit from "date", customer, supplier, part, lineorder
 where lo_custkey = c_custkey and lo_suppkey = s_suppkey and lo_partkey = p_partkey and lo_orderdate = d_datekey
   and c_region = 1 and s_region = 1 and (p_mfgr = 1 or p_mfgr = 2) group by d_year, c_nation order by d_year, c_nation"""


def _date_keys():
    days = np.arange(np.datetime64("1992-01-01"), np.datetime64("1999-01-01"))
    years = days.astype("datetime64[Y]").astype(int) + 1970
    months = days.astype("datetime64[M]").astype(int) % 12 + 1
    dom = (days - days.astype("datetime64[M]")).astype(int) + 1
    return (years * 10000 + months * 100 + dom).astype(np.int32), years.astype(np.int32)


class SsbData:
    """Raw numpy columns.  Codes: p_mfgr 1..5, p_category = mfgr * 10 + 1..5 ('MFGR#12' = 12), p_brand1 = category * 100 + 1..40,
    nation 0..24, region = nation // 5 (AMERICA = 1)."""

    def __init__(self, scale_factor=1.0, seed=7, lineorder_rows=None):
        rng = np.random.default_rng(seed)
        self.scale_factor = scale_factor
        n_customer = max(1, int(30_000 * scale_factor))
        n_supplier = max(1, int(2_000 * scale_factor))
        n_part = max(1, int(200_000 * (math.floor(1 + math.log2(scale_factor)) if scale_factor >= 1 else scale_factor)))
        n_lineorder = lineorder_rows if lineorder_rows is not None else int(6_000_000 * scale_factor)
        self.d_datekey, self.d_year = _date_keys()
        self.c_custkey = np.arange(1, n_customer + 1, dtype=np.int32)
        self.c_nation = rng.integers(0, 25, n_customer, dtype=np.int32)
        self.c_region = (self.c_nation // 5).astype(np.int32)
        self.s_suppkey = np.arange(1, n_supplier + 1, dtype=np.int32)
        self.s_nation = rng.integers(0, 25, n_supplier, dtype=np.int32)
        self.s_region = (self.s_nation // 5).astype(np.int32)
        self.p_partkey = np.arange(1, n_part + 1, dtype=np.int32)
        self.p_mfgr = rng.integers(1, 6, n_part, dtype=np.int32)
        self.p_category = (self.p_mfgr * 10 + rng.integers(1, 6, n_part, dtype=np.int32)).astype(np.int32)
        self.p_brand1 = (self.p_category * 100 + rng.integers(1, 41, n_part, dtype=np.int32)).astype(np.int32)
        self.lo_custkey = rng.integers(1, n_customer + 1, n_lineorder, dtype=np.int32)
        self.lo_partkey = rng.integers(1, n_part + 1, n_lineorder, dtype=np.int32)
        self.lo_suppkey = rng.integers(1, n_supplier + 1, n_lineorder, dtype=np.int32)
        self.lo_orderdate = self.d_datekey[rng.integers(0, len(self.d_datekey), n_lineorder)]
        quantity = rng.integers(1, 51, n_lineorder, dtype=np.int32)
        price = rng.integers(90_000, 200_001, n_lineorder, dtype=np.int32) // 100
        discount = rng.integers(0, 11, n_lineorder, dtype=np.int32)
        self.lo_revenue = (quantity * price * (100 - discount) // 100).astype(np.int32)
        self.lo_supplycost = (price * 6 // 10).astype(np.int32)
        self.n_lineorder = n_lineorder

    TABLES = {"lineorder": ("lo_custkey", "lo_partkey", "lo_suppkey", "lo_orderdate", "lo_revenue", "lo_supplycost"),
              "date": ("d_datekey", "d_year"), "customer": ("c_custkey", "c_nation", "c_region"),
              "supplier": ("s_suppkey", "s_nation", "s_region"), "part": ("p_partkey", "p_mfgr", "p_category", "p_brand1")}

    def host_columns(self, chunk_size=abi.CHUNK_DEFAULT_SIZE):
        """Encoded like Hyrise's automatic encoding: unique keys unencoded, other int columns FrameOfReference
        (segment_encoding_utils.cpp:105-115)."""
        unique = {"c_custkey", "s_suppkey", "p_partkey", "d_datekey"}
        out = {}
        for table, names in self.TABLES.items():
            for name in names:
                encoding = abi.ENC_UNENCODED if name in unique else abi.ENC_FRAME_OF_REFERENCE
                out[name] = storage.make_column(getattr(self, name), None, encoding, chunk_size)
        return out

    def sqlite_result(self, sql):
        """The query's rows from SQLite over the same tables (the reference's own verification practice)."""
        import sqlite3
        db = getattr(self, "_sqlite", None)
        if db is None:   # (loaded once per data set: at scale factor 1 the inserts take half a minute)
            db = sqlite3.connect(":memory:")
            for table, names in self.TABLES.items():
                db.execute(f'create table "{table}" ({", ".join(n + " integer" for n in names)})')
                rows = list(zip(*[getattr(self, n).tolist() for n in names]))
                db.executemany(f'insert into "{table}" values ({", ".join("?" for _ in names)})', rows)
            self._sqlite = db
        return [tuple(r) for r in db.execute(sql).fetchall()]


# ---- the plans ---------------------------------------------------------------------------------------------------------
def _dimension(ex, columns, key, filter_column, condition, value, value2=None):
    """Scan a dimension table, return (its filtered key column = the build side, in chunks of DENSE_CHUNK rows; base RowIDs of the rows)."""
    from .operators import make_predicate
    predicate = make_predicate(condition, abi.TYPE_INT, value, value2)
    rows = ex.scan(columns[filter_column], predicate)
    keys, _ = ex.export(ex.reference_column(columns[key], rows, VIRTUAL_CHUNK), with_nulls=False)   # (materialised: see _fact_key_column)
    return ex.value_column(keys, DENSE_CHUNK), rows


def _join_dimension(ex, build_column, fact_key_column, carried, probe_chunk=VIRTUAL_CHUNK):
    """fact table (as it stands: `carried` = {name: base RowIDs per surviving row}, or None for the base table) joined with one
    filtered dimension.  Returns (positions in the dimension's filtered table, carried RowID arrays of the join's output).
    probe_chunk: the chunk size the probe column presents the join result so far in (its positions are RowIDs of that table)."""
    build_pos, probe_pos = ex.join(build_column, fact_key_column, abi.JOIN_INNER)   # the dimension is the smaller side: build = left
    if carried is None:
        return build_pos, {"lineorder": probe_pos}
    return build_pos, {name: ex.gather_row_ids(rows, probe_chunk, probe_pos) for name, rows in carried.items()}


def _fact_key_column(ex, columns, name, carried):
    """The foreign key `name` of the join result so far as a probe column.  The base table's column before the first join; afterwards
    the key is MATERIALISED (JoinHash materialises its inputs anyway, join_hash_steps.hpp:274-330: here once, by hy_column_export, into a
    plain int32 column the primary-key / foreign-key kernels read with wide loads) instead of being gathered through the PosList by
    both probe passes: SSB Q4.1's second join 1.24 -> 0.45 ms for 36 M rows.  -> (column, chunk size of its positions)"""
    if carried is None:
        return columns[name], VIRTUAL_CHUNK
    through = ex.reference_column(columns[name], carried["lineorder"], VIRTUAL_CHUNK)
    values, _ = ex.export(through, with_nulls=False)
    return ex.value_column(values, DENSE_CHUNK), DENSE_CHUNK


def _join_dimension_repartitioned(comm, ex, key_column, dimension_rows, fact_key_column, carried):
    """The same join as a HASH-REPARTITIONED one (configs[4]: "hash repartition over xGMI"): rank r builds over the filtered dimension
    rows whose key hashes to r -- every rank holds the dimension tables, so its partition is a local cut of the scan's output -- every
    (key, position) of this rank's fact table travels to the rank that owns the key (one all-to-all of keys, one of positions:
    hy_repartition_pack groups them by destination on the device), is joined there against 1 / G of the build side, and every match
    travels back to the rank its fact row came from together with the dimension row it met (the second all-to-all pair).  Returns
    (base RowIDs of the dimension rows, carried RowID arrays) like _join_dimension; pair ORDER differs from the single-process join
    (the aggregate above does not depend on it)."""
    from .distributed import REPARTITION_CHUNK
    torch = comm.torch
    world, me = comm.world, comm.rank
    build_table = ex.reference_column(key_column, dimension_rows, VIRTUAL_CHUNK)
    keys, positions, counts = ex.repartition(build_table, world, 0)
    begin = sum(counts[:me])
    build_keys = keys[begin:begin + counts[me]]
    build_rows = ex.gather_row_ids(dimension_rows, VIRTUAL_CHUNK, positions[begin:begin + counts[me]])   # base RowIDs of this rank's partition
    fact_keys, fact_positions, fact_counts = ex.repartition(fact_key_column, world, 0)
    received_keys, received_counts = comm.all_to_all_var(fact_keys, fact_counts)
    received_positions, _ = comm.all_to_all_var(fact_positions, fact_counts)
    build_pos, probe_pos = ex.join(ex.value_column(build_keys, REPARTITION_CHUNK), ex.value_column(received_keys, REPARTITION_CHUNK), abi.JOIN_INNER)
    matched_dimension = ex.gather_row_ids(build_rows, REPARTITION_CHUNK, build_pos)
    matched_fact = ex.gather_row_ids(received_positions, REPARTITION_CHUNK, probe_pos)
    # the way back: a received tuple's source is the block of the receive buffer it sits in
    flat = probe_pos[:, 0].to(torch.int64) * REPARTITION_CHUNK + probe_pos[:, 1].to(torch.int64)
    bounds = torch.cumsum(torch.tensor(received_counts, dtype=torch.int64, device=flat.device), 0)
    source = torch.bucketize(flat, bounds, right=True)
    order = torch.argsort(source, stable=True)
    back_counts = [int(c) for c in torch.bincount(source, minlength=world).cpu().tolist()]
    dimension_home, _ = comm.all_to_all_var(matched_dimension[order].contiguous(), back_counts)
    fact_home, _ = comm.all_to_all_var(matched_fact[order].contiguous(), back_counts)
    if carried is None:
        return dimension_home, {"lineorder": fact_home}
    return dimension_home, {name: ex.gather_row_ids(rows, VIRTUAL_CHUNK, fact_home) for name, rows in carried.items()}


def run_query(ex, columns, query, fact_first_chunk=0, comm=None, repartitioned=()):
    """columns: {name: executor column} (the fact table's may be one rank's chunk range).  Returns (group-by columns, aggregates)
    ready for ex.aggregate / sharded_aggregate, as executor columns over the join result, plus the number of joined rows.
    comm + repartitioned: the dimensions in `repartitioned` ("customer", "part": the large ones) are joined by hash repartition
    (_join_dimension_repartitioned) instead of against a full replica."""
    if query == "2.1":
        dims = [("part", "p_partkey", "p_category", abi.PRED_EQUALS, 12, None, "lo_partkey"),
                ("supplier", "s_suppkey", "s_region", abi.PRED_EQUALS, AMERICA, None, "lo_suppkey")]
    elif query == "4.1":
        dims = [("supplier", "s_suppkey", "s_region", abi.PRED_EQUALS, AMERICA, None, "lo_suppkey"),
                ("customer", "c_custkey", "c_region", abi.PRED_EQUALS, AMERICA, None, "lo_custkey"),
                ("part", "p_partkey", "p_mfgr", abi.PRED_BETWEEN_INCLUSIVE, 1, 2, "lo_partkey")]   # 'MFGR#1' or 'MFGR#2': the two smallest manufacturers
    else:
        raise ValueError(query)
    carried = None      # base RowIDs per surviving row, per table joined so far
    for table, key, filter_column, condition, value, value2, fact_key in dims:
        build, dimension_rows = _dimension(ex, columns, key, filter_column, condition, value, value2)
        if comm is not None and table in repartitioned:
            fact_column = columns[fact_key] if carried is None else ex.reference_column(columns[fact_key], carried["lineorder"], VIRTUAL_CHUNK)
            carried_dimension, carried = _join_dimension_repartitioned(comm, ex, columns[key], dimension_rows, fact_column, carried)
            carried[table] = carried_dimension
            continue
        fact_column, probe_chunk = _fact_key_column(ex, columns, fact_key, carried)
        build_pos, carried = _join_dimension(ex, build, fact_column, carried, probe_chunk)
        carried[table] = ex.gather_row_ids(dimension_rows, DENSE_CHUNK, build_pos)
    # the date dimension is not filtered: build = the whole d_datekey column
    fact_column, probe_chunk = _fact_key_column(ex, columns, "lo_orderdate", carried)
    date_pos, carried = _join_dimension(ex, columns["d_datekey"], fact_column, carried, probe_chunk)
    carried["date"] = date_pos
    joined = int(date_pos.shape[0])

    def through(name, table):   # a column of the join result, materialised like the foreign keys above (SSB has no NULLs): the aggregate
        values, _ = ex.export(ex.reference_column(columns[name], carried[table], VIRTUAL_CHUNK), with_nulls=False)   # and the projection
        return ex.value_column(values, DENSE_CHUNK)                                                                  # read plain columns

    if query == "2.1":
        return [through("d_year", "date"), through("p_brand1", "part")], [(abi.AGG_SUM, through("lo_revenue", "lineorder"))], joined
    profit = ex.projection(abi.ARITH_SUB, through("lo_revenue", "lineorder"), through("lo_supplycost", "lineorder"))
    return [through("d_year", "date"), through("c_nation", "customer")], [(abi.AGG_SUM, profit)], joined


def star_plan(columns, query):
    """The same two queries as arguments of hy_star_join_aggregate (operators.star_join_aggregate): the plan of run_query + the aggregate
    made by ONE call of the library (csrc/plan.hip) -- single GPU, every dimension against its full replica."""
    from .operators import make_predicate
    if query == "2.1":
        dimensions = [(columns["p_partkey"], columns["p_category"], make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, 12), columns["lo_partkey"]),
                      (columns["s_suppkey"], columns["s_region"], make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, AMERICA), columns["lo_suppkey"]),
                      (columns["d_datekey"], None, None, columns["lo_orderdate"])]
        groupby = [(3, columns["d_year"]), (1, columns["p_brand1"])]
        return dimensions, groupby, [(abi.AGG_SUM, (0, columns["lo_revenue"]), None, None)] + [(abi.AGG_MIN, g, None, None) for g in groupby]
    if query == "4.1":
        dimensions = [(columns["s_suppkey"], columns["s_region"], make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, AMERICA), columns["lo_suppkey"]),
                      (columns["c_custkey"], columns["c_region"], make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, AMERICA), columns["lo_custkey"]),
                      (columns["p_partkey"], columns["p_mfgr"], make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_INT, 1, 2), columns["lo_partkey"]),
                      (columns["d_datekey"], None, None, columns["lo_orderdate"])]
        groupby = [(4, columns["d_year"]), (2, columns["c_nation"])]
        return dimensions, groupby, [(abi.AGG_SUM, (0, columns["lo_revenue"]), abi.ARITH_SUB, (0, columns["lo_supplycost"]))] + [(abi.AGG_MIN, g, None, None) for g in groupby]
    raise ValueError(query)


def star_groups(result, n_groupby_columns):
    """hy_star_join_aggregate's result in the shape aggregate_groups returns: [(key tuple, [aggregate cells])].  The result names
    representative rows of an intermediate table the caller never sees, so star_plan asks for the GROUP BY values as MIN() aggregates of
    the columns themselves (the last n_groupby_columns result columns)."""
    n_cells = len(result.raw) - n_groupby_columns
    keys = [result.column(n_cells + g) for g in range(n_groupby_columns)]
    cells = [result.column(a) for a in range(n_cells)]
    return [(tuple(int(k[i]) for k in keys), [c[i] for c in cells]) for i in range(result.n_groups)]


def sharded_star_groups(comm, columns, query, result=None):
    """N > 1: every rank runs hy_star_join_aggregate over ITS chunks of the fact table (the dimensions are replicated), the ranks' partial
    groups -- a key tuple and the partial SUM each -- are all-gathered and added up per key: integer sums, exact in any order.
    -> ([(key tuple, [sum])] as aggregate_groups returns it, this rank's joined rows, the result object for the next call)"""
    import numpy as np
    from .operators import star_join_aggregate
    torch = comm.torch
    dimensions, groupby, aggregates = star_plan(columns, query)
    result, joined = star_join_aggregate(dimensions, groupby, aggregates, result=result)
    n, n_keys = result.n_groups, len(groupby)
    cells = np.zeros((n, 1 + n_keys), dtype=np.int64)
    if n:
        cells[:, 0] = np.frombuffer(result.raw[0].tobytes(), dtype=np.int64)[:n]                      # SUM: int64
        for g in range(n_keys):
            cells[:, 1 + g] = np.frombuffer(result.raw[1 + g].tobytes(), dtype=np.int32)[:n]          # MIN of an int32 GROUP BY column
    totals = {}
    for part in comm.all_gather_var(torch.from_numpy(cells)):
        for row in part.numpy().tolist():
            key = tuple(row[1:])
            totals[key] = totals.get(key, 0) + row[0]
    return [(key, [total]) for key, total in totals.items()], joined, result


def referenced_bytes(data, query):
    """Algorithmic bytes: referenced lineorder columns x 4 B x N + dimension keys + filter columns."""
    fact = {"2.1": 4, "4.1": 6}[query]   # lo_partkey, lo_suppkey, lo_orderdate, lo_revenue (+ lo_custkey, lo_supplycost)
    dims = {"2.1": len(data.p_partkey) * 2 + len(data.s_suppkey) * 2 + len(data.d_datekey) * 2,
            "4.1": len(data.p_partkey) * 2 + len(data.s_suppkey) * 2 + len(data.c_custkey) * 3 + len(data.d_datekey) * 2}[query]
    return 4 * (fact * data.n_lineorder + dims)


def result_rows(groups):
    """[(key tuple, [cells])] or an aggregate result + key decoder -> sorted [(sum, d_year, brand)] like the SQL's ORDER BY."""
    return sorted((tuple(key), cells[0]) for key, cells in groups)


# ---- measurement (bench.py, tools/ssb_bench.py) -----------------------------------------------------------------------------
def bench(sf=30.0, steps=3, world=1, rank=0, dist=None, share_gpu=False, local_rank=0, verify=False):
    """Q2.1 and Q4.1 at `sf` on this rank's GPU (world > 1: lineorder chunk-sharded, groups combined by the sharded AggregateHash; timed
    twice: every dimension replicated, and `customer` / `part` joined by hash repartition); times are per query, maximum over the
    ranks.  Returns a dict (meaningful on rank 0)."""
    import time
    import torch
    from .distributed import Comm, HipExecutor, aggregate_groups, shard_column, sharded_aggregate
    device = torch.device("cuda", local_rank)
    lib = abi.load_library()
    abi.check(lib.hy_init(local_rank))
    ex = HipExecutor(device)
    t0 = time.perf_counter()
    data = SsbData(scale_factor=sf, seed=7)
    host = data.host_columns()
    t_generate = time.perf_counter() - t0
    fact = set(SsbData.TABLES["lineorder"])
    first_chunk = 0
    columns = {}
    for name, column in host.items():
        if name in fact and world > 1:
            column, first_chunk = shard_column(column, world, rank)
        columns[name] = ex.column(column)
    comm = Comm(dist).bind(torch.device("cpu") if share_gpu else device) if dist is not None else None
    out = {"scale_factor": sf, "lineorder_rows": data.n_lineorder, "n_gpus": world, "generate_and_encode_s": t_generate}
    for query in ("2.1", "4.1"):
        holder = {}

        def timed(repartitioned):
            def once():
                groupby, aggregates, joined = run_query(ex, columns, query, comm=comm, repartitioned=repartitioned)
                if comm is None:
                    holder["groups"] = aggregate_groups(ex, groupby, aggregates)
                else:
                    holder["groups"] = sharded_aggregate(comm, ex, groupby, aggregates, first_chunk)
                holder["joined"] = joined

            once()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                once()
            torch.cuda.synchronize()
            seconds = (time.perf_counter() - t0) / steps
            joined = holder["joined"]
            if comm is not None:
                t = torch.tensor([seconds], dtype=torch.float64, device=comm._device)
                comm.all_reduce(t, "max")
                seconds = float(t.item())
                j = torch.tensor([joined], dtype=torch.int64, device=comm._device)
                comm.all_reduce(j, "sum")
                joined = int(j.item())
            return seconds, joined

        seconds, joined = timed(())
        algorithmic = referenced_bytes(data, query)
        entry = {"ms": seconds * 1e3, "lineorder_rows_per_s": data.n_lineorder / seconds, "joined_rows": joined, "groups": len(holder["groups"]),
                 "algorithmic_bytes": algorithmic, "GBps_on_algorithmic_bytes": algorithmic / seconds / 1e9}
        if comm is None:   # the same plan made by ONE call of the library (hy_star_join_aggregate): no interpreter between the operator calls
            from .operators import star_join_aggregate
            dimensions, star_groupby, star_aggregates = star_plan(columns, query)
            star = {}

            def one_call():
                star["result"], star["joined"] = star_join_aggregate(dimensions, star_groupby, star_aggregates, result=star.get("result"))

            one_call()
            if result_rows(star_groups(star["result"], len(star_groupby))) != result_rows(holder["groups"]) or star["joined"] != joined:
                raise RuntimeError(f"SSB Q{query}: hy_star_join_aggregate and the operator chain disagree")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_call()
            torch.cuda.synchronize()
            star_seconds = (time.perf_counter() - t0) / steps
            entry["operator_calls_from_python_ms"] = entry["ms"]
            entry.update({"ms": star_seconds * 1e3, "lineorder_rows_per_s": data.n_lineorder / star_seconds, "GBps_on_algorithmic_bytes": algorithmic / star_seconds / 1e9,
                          "plan": "hy_star_join_aggregate: scan -> JoinHash per dimension -> projection -> AggregateHash as one call of the library (csrc/plan.hip, join_star.hpp: "
                                  "every dimension probed in one pass over lineorder, the survivors grouped inside it); "
                                  "operator_calls_from_python_ms: the same calls made one by one through ctypes (hyrise_amd/ssb.py run_query)"})
        if comm is not None:   # every rank's shard as ONE hy_star_join_aggregate call, the ranks' partial groups added up
            star = {}

            def one_call_per_rank():
                star["groups"], star["joined"], star["result"] = sharded_star_groups(comm, columns, query, star.get("result"))

            one_call_per_rank()
            if result_rows(star["groups"]) != result_rows(holder["groups"]):
                raise RuntimeError(f"SSB Q{query}: the ranks' hy_star_join_aggregate results and the operator chain disagree")
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_call_per_rank()
            torch.cuda.synchronize()
            star_seconds = (time.perf_counter() - t0) / steps
            t = torch.tensor([star_seconds], dtype=torch.float64, device=comm._device)
            comm.all_reduce(t, "max")
            star_seconds = float(t.item())
            entry["operator_calls_from_python_ms"] = entry["ms"]
            entry.update({"ms": star_seconds * 1e3, "lineorder_rows_per_s": data.n_lineorder / star_seconds, "GBps_on_algorithmic_bytes": algorithmic / star_seconds / 1e9})
        if comm is not None:   # the same query with `customer` / `part` joined by hash repartition (tuples to the key's rank and back)
            entry["plan"] = "lineorder chunk-sharded, dimensions replicated: hy_star_join_aggregate per rank, the partial groups all-gathered and added (operator_calls_from_python_ms: the operator chain per rank, groups all-reduced)"
            try:
                repartition_seconds, repartition_joined = timed(("customer", "part"))
                entry["repartitioned"] = {"plan": "customer and part joined by hash repartition (two all-to-all pairs per join), the other dimensions replicated",
                                          "ms": repartition_seconds * 1e3, "lineorder_rows_per_s": data.n_lineorder / repartition_seconds,
                                          "joined_rows": repartition_joined, "groups": len(holder["groups"])}
            except Exception as error:   # (reported, not fatal: the replicated plan's numbers above stand on their own)
                entry["repartitioned"] = {"error": f"{type(error).__name__}: {error}"}
        if verify and rank == 0:
            sql = Q2_1_SQL if query == "2.1" else Q4_1_SQL
            rows = data.sqlite_result(sql)
            want = sorted(((r[1], r[2]), r[0]) for r in rows) if query == "2.1" else sorted(((r[0], r[1]), r[2]) for r in rows)
            entry["matches_sqlite"] = result_rows(holder["groups"]) == want
        entry["_rows"] = result_rows(holder["groups"])   # (for the caller's parity check against the CPU oracle; bench.py drops it from its output)
        out[f"q{query}"] = entry
    return out


