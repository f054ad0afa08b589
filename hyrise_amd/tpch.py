"""TPC-H lineitem / orders shaped synthetic columns (numpy, seeded) -- the inputs of bench.py and of the large tests.

The reference generates TPC-H with its vendored dbgen (src/benchmarklib/tpch/tpch_table_generator.cpp:141-316), which
cannot travel to the GPU box; this module follows the TPC-H specification's column distributions (the same ones
dbgen implements) so that value ranges, dictionary sizes and vector-compression widths match what Hyrise would hold:
  o_orderkey   sparse keys: 8 of every 32 integers (dbgen MK_SPARSE), ascending, unique   -> unencoded int32
  o_orderdate  uniform in [1992-01-01, 1998-08-02]
  lineitems    1..7 per order (mean 4)                                                     -> l_orderkey FoR, u16 offsets
  l_shipdate   o_orderdate + uniform[1,121]; l_commitdate +[30,90]; l_receiptdate = ship + [1,30]
  l_quantity   uniform[1,50]; l_discount uniform[0.00,0.10]; l_tax uniform[0.00,0.08]
  l_extendedprice  quantity * retail price of a uniformly chosen part
  l_returnflag 'R'/'A' if receiptdate <= 1995-06-17 else 'N'; l_linestatus 'O' if shipdate > 1995-06-17 else 'F'
Dates are int32 days since 1992-01-01 (the "int" twin of SURVEY.md section 8: a dictionary scan only ever sees value
ids, so DictionarySegment<int32> of day numbers and DictionarySegment<pmr_string> of ISO dates run the same kernel).
"""
import os

import numpy as np

from . import abi, storage

ORDERS_PER_SF = 1_500_000
LINEITEM_ROWS_SF10 = 59_986_052          # dbgen row count at SF 10 (tpch_table_generator.cpp:155-166)
LAST_ORDERDATE = 2405                    # 1998-08-02 as days since 1992-01-01
CURRENT_DATE = 1263                      # 1995-06-17
DAY_1995_01_01 = 1096
DAY_1994_01_01 = 731
DAY_1998_09_02 = 2436


def sparse_orderkeys(n):
    """dbgen mk_sparse: ((i >> 3) << 5) | (i & 7) for i = 1..n."""
    i = np.arange(1, n + 1, dtype=np.int64)
    return (((i >> 3) << 5) | (i & 7)).astype(np.int32)


class TpchData:
    """Raw (unencoded) numpy columns of orders and lineitem at `scale_factor`."""

    def __init__(self, scale_factor=10.0, seed=42, lineitem_rows=None, keys_only=False):
        # keys_only: o_orderkey and l_orderkey alone (the join's inputs; the same keys the full generator produces)
        # HY_TPCH_CACHE=<directory>: the profiling scripts run the same generator many times in one session
        self.keys_only = keys_only
        cache = os.environ.get("HY_TPCH_CACHE")
        path = os.path.join(cache, f"tpch_{scale_factor}_{seed}_{lineitem_rows}{'_keys' if keys_only else ''}.npz") if cache else None
        if path and os.path.exists(path):
            with np.load(path) as arrays:
                for name in arrays.files:
                    setattr(self, name, arrays[name])
            self.n_orders, self.n_lineitems = len(self.o_orderkey), len(self.l_orderkey)
            return
        self._generate(scale_factor, seed, lineitem_rows)
        if path:
            os.makedirs(cache, exist_ok=True)
            np.savez(path, **{name: value for name, value in vars(self).items() if isinstance(value, np.ndarray)})

    def _generate(self, scale_factor, seed, lineitem_rows):
        rng = np.random.default_rng(seed)
        n_orders = int(round(ORDERS_PER_SF * scale_factor))
        self.o_orderkey = sparse_orderkeys(n_orders)
        self.o_orderdate = rng.integers(0, LAST_ORDERDATE + 1, n_orders, dtype=np.int32)
        counts = rng.integers(1, 8, n_orders, dtype=np.int32)
        if lineitem_rows is None and abs(scale_factor - 10.0) < 1e-9:
            lineitem_rows = LINEITEM_ROWS_SF10
        if lineitem_rows is not None:   # pin the row count (dbgen's differs from ours only by its fixed seeds)
            diff = int(counts.sum()) - lineitem_rows
            idx = 0
            while diff != 0:
                c = counts[idx]
                if diff > 0 and c > 1:
                    step = min(diff, c - 1)
                    counts[idx] -= step
                    diff -= step
                elif diff < 0 and c < 7:
                    step = min(-diff, 7 - c)
                    counts[idx] += step
                    diff += step
                idx += 1
        self.lineitems_per_order = counts
        n = int(counts.sum())
        order_index = np.repeat(np.arange(n_orders, dtype=np.int32), counts)
        self.l_orderkey = self.o_orderkey[order_index]
        self.n_orders, self.n_lineitems = n_orders, n
        if self.keys_only:
            return
        orderdate = self.o_orderdate[order_index]
        self.l_shipdate = (orderdate + rng.integers(1, 122, n, dtype=np.int32)).astype(np.int32)
        self.l_commitdate = (orderdate + rng.integers(30, 91, n, dtype=np.int32)).astype(np.int32)
        self.l_receiptdate = (self.l_shipdate + rng.integers(1, 31, n, dtype=np.int32)).astype(np.int32)
        self.l_quantity = rng.integers(1, 51, n).astype(np.float32)
        self.l_discount = (rng.integers(0, 11, n) / 100.0).astype(np.float32)
        self.l_tax = (rng.integers(0, 9, n) / 100.0).astype(np.float32)
        partkey = rng.integers(1, int(200_000 * scale_factor) + 1, n)
        retail = (90000 + (partkey // 10) % 20001 + 100 * (partkey % 1000)) / 100.0
        self.l_extendedprice = (self.l_quantity * retail).astype(np.float32)
        returned = rng.integers(0, 2, n).astype(np.uint8)
        # single-character strings as their byte (the AggregateHash key of a <5-char string is 2 + byte,
        # aggregate_hash.cpp:852-900)
        self.l_returnflag = np.where(self.l_receiptdate <= CURRENT_DATE, np.where(returned == 1, ord("R"), ord("A")),
                                     ord("N")).astype(np.int32)
        self.l_linestatus = np.where(self.l_shipdate > CURRENT_DATE, ord("O"), ord("F")).astype(np.int32)
        self.n_orders, self.n_lineitems = n_orders, n


def shipdate_column(n_rows=LINEITEM_ROWS_SF10, seed=42, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    """Just l_shipdate (config 2), dictionary-encoded per chunk like Hyrise's "Automatic" encoding of that column."""
    rng = np.random.default_rng(seed)
    days = (rng.integers(0, LAST_ORDERDATE + 1, n_rows, dtype=np.int32) + rng.integers(1, 122, n_rows, dtype=np.int32))
    return days.astype(np.int32), storage.make_column(days.astype(np.int32), None, abi.ENC_DICTIONARY, chunk_size)


def iso_date(day):
    """Day number (days since 1992-01-01) -> b'YYYY-MM-DD', the form TPC-H dates have in Hyrise's string columns."""
    import datetime
    return (datetime.date(1992, 1, 1) + datetime.timedelta(days=int(day))).isoformat().encode("ascii")


def string_date_column(int_column):
    """The DictionarySegment<pmr_string> twin of a dictionary-encoded column of day numbers -- what l_shipdate IS in Hyrise's
    TPC-H schema (tpch_table_generator.cpp:46).  ISO dates sort like the days they name, so every chunk's attribute vector is the
    int twin's, and its dictionary (kept on the host, like all string dictionaries) holds the same days as strings.
    -> (HostColumn of type string, per-chunk dictionaries as sorted lists of bytes)"""
    segments, dictionaries = [], []
    for s in int_column.segments:
        assert s.encoding == abi.ENC_DICTIONARY
        dictionaries.append([iso_date(day) for day in np.asarray(s.aux)[:s.aux_size]])
        segments.append(storage.HostSegment(abi.ENC_DICTIONARY, abi.TYPE_STRING, s.size, s.width, s.data, aux=None, aux_size=s.aux_size))
    return storage.HostColumn(segments, abi.TYPE_STRING), dictionaries


def string_key_column(char_codes, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    """A column of one-character strings (l_returnflag, l_linestatus) as AggregateHash's GROUP BY takes it from the adapter:
    DictionarySegment<pmr_string> per chunk -- attribute vector of value ids (u8: a handful of distinct flags) over the
    byte-ordered dictionary -- whose dictionary entries are replaced by their AggregateKeyEntry names, 2 + the character
    (aggregate_hash.cpp:852-900, hyrise_amd/string_keys.py).  `char_codes`: the characters as integers."""
    column = storage.make_column(np.ascontiguousarray(char_codes, dtype=np.int32), None, abi.ENC_DICTIONARY, chunk_size)
    segments = [storage.HostSegment(abi.ENC_DICTIONARY, abi.TYPE_LONG, s.size, s.width, s.data, aux=(s.aux.astype(np.int64) + 2), aux_size=s.aux_size)
                for s in column.segments]
    return storage.HostColumn(segments, abi.TYPE_LONG)


def q1_core_columns(data, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    """Config 4 as SURVEY.md 8(d) specifies it: GROUP BY l_returnflag, l_linestatus (string dictionaries, u8 value ids) and
    the measures as DictionarySegment<float> (l_quantity u8, l_extendedprice u16 with ~60 k entries per chunk, l_discount u8).
    Returns (group-by columns, {name: column}, algorithmic bytes: attribute vectors + dictionaries, each read once)."""
    groupby = [string_key_column(data.l_returnflag, chunk_size), string_key_column(data.l_linestatus, chunk_size)]
    measures = {name: storage.make_column(getattr(data, name), None, abi.ENC_DICTIONARY, chunk_size) for name in ("l_quantity", "l_extendedprice", "l_discount")}
    total = 0
    for column in groupby + list(measures.values()):
        for s in column.segments:
            total += s.size * s.width + s.aux_size * s.aux.dtype.itemsize
    return groupby, measures, total


# ---- TPC-H Q6 as an operator chain (configs[0]: the reference's own CPU-runnable case) -------------------------------------
Q6_SQL = ("SELECT SUM(l_extendedprice * l_discount) AS revenue, COUNT(*) FROM lineitem WHERE l_shipdate >= :from AND l_shipdate < :to "
          "AND l_discount BETWEEN :low AND :high AND l_quantity < :quantity")


def q6_columns(data, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    """The four lineitem columns Q6 touches: l_shipdate / l_discount dictionary-encoded, l_quantity / l_extendedprice as
    float value segments."""
    return {"l_shipdate": storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY, chunk_size),
            "l_discount": storage.make_column(data.l_discount, None, abi.ENC_DICTIONARY, chunk_size),
            "l_quantity": storage.make_column(data.l_quantity, None, abi.ENC_UNENCODED, chunk_size),
            "l_extendedprice": storage.make_column(data.l_extendedprice, None, abi.ENC_UNENCODED, chunk_size)}


def run_q6(ex, columns, date_from=DAY_1994_01_01, date_to=DAY_1995_01_01, discount=(0.05, 0.07), quantity=24.0, mvcc=None, transaction=(0, 0)):
    """TPC-H Q6 (tpch_queries.cpp:206-210) the way the reference's plan runs it: three TableScans chained through reference
    segments (the second and third scan read the PosList of the one before, table_scan.cpp:158-196), a Projection
    l_extendedprice * l_discount over the survivors, AggregateHash SUM without GROUP BY.  `ex`: distributed.HipExecutor (every
    intermediate -- PosLists, the product column -- stays in device memory) or the tests' oracle executor.
    With `mvcc` (the table's MvccData as a column, storage.make_mvcc_column) the chain starts like every SQL-driven plan of the
    reference: GetTable -> Validate -> the scans (sql_pipeline_builder.hpp:55, validate.cpp:275); `transaction` = (our tid,
    snapshot commit id).
    -> (revenue, qualifying rows)"""
    from .operators import make_predicate
    # the scans keep the reference's output shape -- one PosList per input chunk, each referencing one data chunk -- so the
    # second and third scan take the single-chunk path of AbstractDereferencedColumnTableScanImpl (:38-46)
    first = columns["l_shipdate"]
    if mvcc is not None:
        first = ex.reference_column_chunked(first, ex.validate_chunked(mvcc, transaction[0], transaction[1]))
    lists = ex.scan_chunked(first, make_predicate(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, date_from, date_to))
    lists = ex.scan_chunked(ex.reference_column_chunked(columns["l_discount"], lists),
                            make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_FLOAT, np.float32(discount[0]), np.float32(discount[1])))
    lists = ex.scan_chunked(ex.reference_column_chunked(columns["l_quantity"], lists), make_predicate(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, quantity))
    if lists.total == 0:
        return None, 0        # SUM over no rows is NULL
    revenue = ex.projection(abi.ARITH_MUL, ex.reference_column_chunked(columns["l_extendedprice"], lists), ex.reference_column_chunked(columns["l_discount"], lists))
    result = ex.aggregate([], [(abi.AGG_SUM, revenue), (abi.AGG_COUNT, None)])
    return result.column(0)[0], result.column(1)[0]


def q6_fused(columns, date_from=DAY_1994_01_01, date_to=DAY_1995_01_01, discount=(0.05, 0.07), quantity=24.0, mvcc=None, transaction=(0, 0)):
    """The same Q6 in ONE pass over lineitem (hy_scan_project_aggregate): no PosList, no product column.  `columns`: DeviceColumns of the
    data table; `mvcc` (a DeviceColumn of the table's MvccData): Validate is the pass's first filter, like in run_q6.
    -> (revenue, qualifying rows)"""
    from .operators import make_predicate, scan_project_aggregate, validate_filter
    result = scan_project_aggregate(
        ([validate_filter(mvcc, transaction[0], transaction[1])] if mvcc is not None else []) +
        [(columns["l_shipdate"], make_predicate(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, date_from, date_to)),
         (columns["l_discount"], make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_FLOAT, np.float32(discount[0]), np.float32(discount[1]))),
         (columns["l_quantity"], make_predicate(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, quantity))],
        [], [(abi.AGG_SUM, (abi.ARITH_MUL, columns["l_extendedprice"], columns["l_discount"])), (abi.AGG_COUNT, None)], group_capacity=4)
    return result.column(0)[0], result.column(1)[0]


# ---- TPC-H Q1, the whole query (tpch_queries.cpp:60-80): scan, two expressions, eight aggregates ---------------------------------------
Q1_AGGREGATES = ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order")


def q1_columns(data, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    """The seven lineitem columns Q1 reads, encoded like q1_core_columns (+ l_shipdate as a dictionary of days, l_tax)."""
    groupby, measures, _ = q1_core_columns(data, chunk_size)
    columns = dict(measures)
    columns["l_returnflag"], columns["l_linestatus"] = groupby
    columns["l_shipdate"] = storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY, chunk_size)
    columns["l_tax"] = storage.make_column(data.l_tax, None, abi.ENC_DICTIONARY, chunk_size)
    return columns


def _q1_aggregates(quantity, price, discount, disc_price, charge):
    return [(abi.AGG_SUM, quantity), (abi.AGG_SUM, price), (abi.AGG_SUM, disc_price), (abi.AGG_SUM, charge), (abi.AGG_AVG, quantity), (abi.AGG_AVG, price),
            (abi.AGG_AVG, discount), (abi.AGG_COUNT, None)]


def run_q1(ex, columns, ship_to=DAY_1998_09_02):
    """Q1 the way the reference's plan runs it: TableScan l_shipdate <= :to, Projection of l_extendedprice * (1 - l_discount) and of
    that * (1 + l_tax) over the survivors (four ArithmeticExpressions, each materialised), AggregateHash GROUP BY l_returnflag,
    l_linestatus.  -> aggregate result (the ORDER BY is not part of the path)"""
    from .operators import make_predicate
    lists = ex.scan_chunked(columns["l_shipdate"], make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, ship_to))
    ref = {name: ex.reference_column_chunked(column, lists) for name, column in columns.items() if name != "l_shipdate"}
    one = (abi.TYPE_INT, 1)
    disc_price = ex.projection(abi.ARITH_MUL, ref["l_extendedprice"], ex.projection(abi.ARITH_SUB, one, ref["l_discount"]))
    charge = ex.projection(abi.ARITH_MUL, disc_price, ex.projection(abi.ARITH_ADD, one, ref["l_tax"]))
    return ex.aggregate([ref["l_returnflag"], ref["l_linestatus"]], _q1_aggregates(ref["l_quantity"], ref["l_extendedprice"], ref["l_discount"], disc_price, charge))


def q1_fused(columns, ship_to=DAY_1998_09_02):
    """The same Q1 in ONE pass over lineitem (hy_scan_project_aggregate)."""
    from .operators import make_predicate, scan_project_aggregate
    one = (abi.TYPE_INT, 1)
    disc_price = (abi.ARITH_MUL, columns["l_extendedprice"], (abi.ARITH_SUB, one, columns["l_discount"]))
    charge = (abi.ARITH_MUL, disc_price, (abi.ARITH_ADD, one, columns["l_tax"]))
    return scan_project_aggregate([(columns["l_shipdate"], make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, ship_to))],
                                  [columns["l_returnflag"], columns["l_linestatus"]],
                                  _q1_aggregates(columns["l_quantity"], columns["l_extendedprice"], columns["l_discount"], disc_price, charge), group_capacity=4096)


# ---- rows of the reference's own generator --------------------------------------------------------------------------------
def convert_money(cents):
    """tpch_table_generator.cpp:90-94: float(dollars) + float(cents) / 100.0f, in float32 arithmetic."""
    cents = np.asarray(cents, dtype=np.int64)
    return (cents // 100).astype(np.float32) + (cents % 100).astype(np.float32) / np.float32(100.0)


def _days_of_iso(dates):
    """char[n][10] 'YYYY-MM-DD' -> int32 days since 1992-01-01"""
    text = np.ascontiguousarray(dates).view("S10").reshape(-1)
    return (text.astype("U10").astype("datetime64[D]") - np.datetime64("1992-01-01")).astype(np.int32)


class DbgenData:
    """orders / lineitem rows as the reference's vendored dbgen generates them (third_party/tpch-dbgen driven like
    TPCHTableGenerator::generate, oracle/dbgen/tpch_rows.c), with TpchData's attributes -- so the same plans, encoders and tests run on
    them -- and in Hyrise's column types (tpch_table_generator.cpp:34-47): keys int32, l_quantity / l_extendedprice / l_discount / l_tax
    float through convert_money, flags and dates strings (dates kept as day numbers; string_date_column makes the string twin).
    Sources: the binary file tpch_rows writes (oracle/_ref: built where the reference tree is), or the committed fixture
    tests/golden/dbgen/*.npz that tools/make_dbgen_fixture.py derives from such a file."""

    def __init__(self, arrays):
        self.o_orderkey = arrays["o_orderkey"].astype(np.int32)
        self.l_orderkey = arrays["l_orderkey"].astype(np.int32)
        self.l_quantity = arrays["l_quantity"].astype(np.float32)
        self.l_extendedprice = convert_money(arrays["l_extendedprice_cents"])
        self.l_discount = convert_money(arrays["l_discount_cents"])
        self.l_tax = convert_money(arrays["l_tax_cents"])
        self.l_returnflag = arrays["l_returnflag"].astype(np.int32)     # the character's code, like TpchData
        self.l_linestatus = arrays["l_linestatus"].astype(np.int32)
        self.l_shipdate = arrays["l_shipdate"].astype(np.int32)
        self.l_commitdate = arrays["l_commitdate"].astype(np.int32)
        self.l_receiptdate = arrays["l_receiptdate"].astype(np.int32)
        self.n_orders, self.n_lineitems = len(self.o_orderkey), len(self.l_orderkey)
        self.keys_only = False

    @staticmethod
    def read_rows_file(path):
        """The arrays of a file written by oracle/_ref/tpch_rows (layout: oracle/dbgen/tpch_rows.c)."""
        with open(path, "rb") as fh:
            assert fh.read(8) == b"HYDBGEN1", "not a tpch_rows file"
            n_orders, n = (int(x) for x in np.frombuffer(fh.read(16), dtype="<u8"))

            def take(dtype, count, width=1):
                return np.frombuffer(fh.read(np.dtype(dtype).itemsize * count * width), dtype=dtype)

            arrays = {"o_orderkey": take("<i4", n_orders), "l_orderkey": take("<i4", n), "l_quantity": take("<i4", n), "l_extendedprice_cents": take("<i8", n),
                      "l_discount_cents": take("<i4", n), "l_tax_cents": take("<i4", n), "l_returnflag": take("u1", n), "l_linestatus": take("u1", n)}
            for name in ("l_shipdate", "l_commitdate", "l_receiptdate"):
                arrays[name] = _days_of_iso(take("u1", n, 10).reshape(n, 10))
        return arrays

    @classmethod
    def from_rows_file(cls, path):
        return cls(cls.read_rows_file(path))

    @classmethod
    def from_fixture(cls, path):
        with np.load(path) as arrays:
            return cls({name: arrays[name] for name in arrays.files})

    @classmethod
    def generate(cls, scale_factor, binary, scratch_directory):
        """Runs the generator (`binary` = oracle/_ref/tpch_rows) at `scale_factor`."""
        import subprocess
        path = os.path.join(scratch_directory, f"tpch_rows_{scale_factor}.bin")
        subprocess.check_call([binary, str(scale_factor), path], stderr=subprocess.DEVNULL)
        try:
            return cls.from_rows_file(path)
        finally:
            os.remove(path)
