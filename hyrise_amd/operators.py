"""Thin Python callers of the C ABI (tests / bench plumbing).  Every function here ends in a call into
libhyrise_amd.so; nothing is computed in Python."""
import ctypes as C

import numpy as np

from . import abi


def make_predicate(condition, data_type=abi.TYPE_INT, value=None, value2=None, nullable=False, per_chunk_lower=None,
                   per_chunk_upper=None, per_chunk_found=None, dictionary_matches=None):
    """dictionary_matches (LIKE family): per data chunk a bool array over the chunk's dictionary -- what
    ColumnLikeTableScanImpl::_find_matches_in_dictionary computes."""
    p = abi.Predicate()
    p.condition = condition
    p.value_type = data_type
    for field, v in (("value", value), ("value2", value2)):
        if v is None:
            continue
        u = getattr(p, field)
        if data_type == abi.TYPE_INT:
            u.i32 = int(v)
        elif data_type == abi.TYPE_LONG:
            u.i64 = int(v)
        elif data_type == abi.TYPE_FLOAT:
            u.f32 = float(v)
        elif data_type == abi.TYPE_DOUBLE:
            u.f64 = float(v)
    p.column_is_nullable = 1 if nullable else 0
    keep = []
    for name, arr, dt in (("per_chunk_lower", per_chunk_lower, np.uint32), ("per_chunk_upper", per_chunk_upper, np.uint32),
                          ("per_chunk_found", per_chunk_found, np.uint8)):
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=dt)
            keep.append(arr)
            setattr(p, name, arr.ctypes.data)
    if dictionary_matches is not None:
        words, offsets = [], [0]
        for matches in dictionary_matches:
            bits = np.zeros(((len(matches) + 63) // 64) * 64, dtype=np.uint8)
            bits[:len(matches)] = np.asarray(matches, dtype=bool)
            words.append(np.packbits(bits, bitorder="little").view(np.uint64))
            offsets.append(offsets[-1] + len(words[-1]))
        flat = np.ascontiguousarray(np.concatenate(words) if words else np.zeros(0, dtype=np.uint64))
        if len(flat) == 0:
            flat = np.zeros(1, dtype=np.uint64)
        starts = np.array(offsets, dtype=np.uint64)
        keep += [flat, starts]
        p.match_words, p.match_word_offsets = flat.ctypes.data, starts.ctypes.data
    p._keepalive = keep
    return p


def string_predicate(condition, dictionaries, value, value2=None, nullable=False):
    """The predicate of `string_column <condition> 'literal'` over DictionarySegment<pmr_string> chunks: the literal(s) are
    resolved against every chunk's (byte-wise sorted) dictionary on the host -- lower_bound / upper_bound, as
    column_vs_value_table_scan_impl.cpp:211-226 and column_between_table_scan_impl.cpp:112-124 do per segment -- and the device
    compares value ids.  `dictionaries`: per chunk the sorted list of bytes."""
    import bisect

    def encoded(x):
        return x if isinstance(x, bytes) else str(x).encode("utf-8")

    first, second = encoded(value), (encoded(value2) if value2 is not None else None)
    between = abi.PRED_BETWEEN_INCLUSIVE <= condition <= abi.PRED_BETWEEN_EXCLUSIVE
    lower_inclusive = condition in (abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_BETWEEN_UPPER_EXCLUSIVE)
    upper_inclusive = condition in (abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_BETWEEN_LOWER_EXCLUSIVE)
    lower, upper, found = [], [], []
    for dictionary in dictionaries:
        def bound(position):
            return position if position < len(dictionary) else abi.INVALID_VALUE_ID
        if between:
            lower.append(bound(bisect.bisect_left(dictionary, first) if lower_inclusive else bisect.bisect_right(dictionary, first)))
            upper.append(bound(bisect.bisect_right(dictionary, second) if upper_inclusive else bisect.bisect_left(dictionary, second)))
            found.append(0)
        else:
            position = bisect.bisect_left(dictionary, first)
            lower.append(bound(position))
            upper.append(bound(bisect.bisect_right(dictionary, first)))
            found.append(1 if position < len(dictionary) and dictionary[position] == first else 0)
    return make_predicate(condition, abi.TYPE_STRING, nullable=nullable, per_chunk_lower=lower, per_chunk_upper=upper, per_chunk_found=found)


def _literal(data_type, value):
    v = abi.Value()
    C.memset(C.byref(v), 0, C.sizeof(v))
    if data_type == abi.TYPE_INT:
        v.i32 = int(value)
    elif data_type == abi.TYPE_LONG:
        v.i64 = int(value)
    elif data_type == abi.TYPE_FLOAT:
        v.f32 = float(value)
    elif data_type == abi.TYPE_DOUBLE:
        v.f64 = float(value)
    return v


def predicate_for_column(condition, column_type, literal_type, literal, literal2_type=None, literal2=None, nullable=False):
    """The predicate of `column <condition> literal [AND literal2]` as the scan takes it: hy_predicate_cast applies
    TableScan::create_impl's lossless predicate cast (table_scan.cpp:336-366, 406-448).  Returns None where the reference
    falls back to its ExpressionEvaluator scan (HY_ERR_UNSUPPORTED)."""
    lib = abi.load_library()
    out = abi.Predicate()
    first = _literal(literal_type, literal)
    second = _literal(literal2_type, literal2) if literal2 is not None else None
    status = lib.hy_predicate_cast(condition, column_type, literal_type, C.addressof(first), literal2_type if literal2 is not None else abi.TYPE_NULL,
                                   C.addressof(second) if second is not None else None, C.byref(out))
    if status == abi.ERR_UNSUPPORTED:
        return None
    abi.check(status)
    out.column_is_nullable = 1 if nullable else 0
    return out


def join_output_chunks(slice_offsets, n_slices):
    """Pair ranges of the output chunks JoinHash builds from a join result's PosLists (hy_join_output_chunks: write_output_chunks'
    1000 / 4000 merge, join_output_writing.cpp:245-296) -> numpy uint64 array of n_chunks + 1 offsets."""
    lib = abi.load_library()
    offsets = np.ascontiguousarray(slice_offsets[:n_slices + 1], dtype=np.uint64)
    out = np.zeros(n_slices + 1, dtype=np.uint64)
    n = C.c_uint32(0)
    abi.check(lib.hy_join_output_chunks(offsets.ctypes.data, n_slices, out.ctypes.data, C.byref(n)))
    return out[:n.value + 1]


class HostScanResult:
    """Scan result in host memory, numpy views."""

    def __init__(self, n_chunks, capacity, flags=0):
        self.matches = np.zeros((max(1, capacity), 2), dtype=np.uint32)
        self.offsets = np.zeros(n_chunks + 1, dtype=np.uint64)
        self.counts = np.zeros(max(1, n_chunks), dtype=np.uint32)
        self.chunk_state = np.zeros(max(1, n_chunks), dtype=np.uint8)
        self.n_chunks = n_chunks
        r = abi.ScanResult()
        r.mem = abi.MEM_HOST
        r.flags = flags
        r.matches = self.matches.ctypes.data
        r.capacity = capacity
        r.offsets = self.offsets.ctypes.data
        r.counts = self.counts.ctypes.data
        r.chunk_state = self.chunk_state.ctypes.data
        self.c = r

    def pos_list(self, chunk):
        return self.matches[int(self.offsets[chunk]):int(self.offsets[chunk + 1])]

    @property
    def total(self):
        return int(self.offsets[self.n_chunks])


def table_scan(column, predicate, excluded_chunks=None, flags=0, capacity=None):
    """hy_table_scan with a host-memory result."""
    lib = abi.load_library()
    result = HostScanResult(column.n_chunks, column.rows if capacity is None else capacity, flags)
    excluded = None
    n_excluded = 0
    if excluded_chunks is not None and len(excluded_chunks):
        excluded = np.ascontiguousarray(excluded_chunks, dtype=np.uint32)
        n_excluded = len(excluded)
    abi.check(lib.hy_table_scan(column.handle, C.byref(predicate), excluded.ctypes.data if excluded is not None else None,
                                n_excluded, C.byref(result.c)))
    return result


def table_scan_columns(left, right, condition, capacity=None):
    lib = abi.load_library()
    result = HostScanResult(left.n_chunks, left.rows if capacity is None else capacity)
    abi.check(lib.hy_table_scan_columns(left.handle, right.handle, condition, C.byref(result.c)))
    return result


def validate(mvcc, our_tid, snapshot_commit_id, can_use_chunk_shortcut=True, flags=0):
    """Validate (MVCC visibility) over a DeviceColumn of MVCC segments or of reference segments into one."""
    lib = abi.load_library()
    result = HostScanResult(mvcc.n_chunks, mvcc.rows, flags)
    abi.check(lib.hy_validate(mvcc.handle, our_tid, snapshot_commit_id, 1 if can_use_chunk_shortcut else 0, C.byref(result.c)))
    return result


class ResultColumn:
    """A device-resident column produced by an operator (hy_projection_arithmetic): usable wherever a DeviceColumn is."""

    def __init__(self, handle):
        self.lib = abi.load_library()
        self.handle = handle
        rows, chunks = C.c_uint64(0), C.c_uint32(0)
        abi.check(self.lib.hy_column_row_count(handle, C.byref(rows)))
        abi.check(self.lib.hy_column_chunk_count(handle, C.byref(chunks)))
        self.rows, self.n_chunks = int(rows.value), int(chunks.value)
        self.data_type = int(self.lib.hy_column_data_type(handle))

    def read(self):
        """(values, null mask) of the whole column, copied back to the host."""
        from .storage import NP_TYPES
        values, nulls = [], []
        for chunk in range(self.n_chunks):
            rows = int(self.lib.hy_column_chunk_rows(self.handle, chunk))
            v = np.zeros(rows, dtype=NP_TYPES[self.data_type])
            words = np.zeros((rows + 63) // 64, dtype=np.uint64)
            abi.check(self.lib.hy_column_read_chunk(self.handle, chunk, v.ctypes.data, words.ctypes.data))
            values.append(v)
            nulls.append(np.unpackbits(words.view(np.uint8), bitorder="little")[:rows].astype(bool))
        if not values:
            return np.zeros(0, dtype=NP_TYPES[self.data_type]), np.zeros(0, dtype=bool)
        return np.concatenate(values), np.concatenate(nulls)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hy_column_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _operand(x):
    """DeviceColumn / ResultColumn, or a literal: (HY_TYPE_*, value) / None for a NULL literal."""
    o = abi.Operand()
    if hasattr(x, "handle"):
        o.column = x.handle
        return o
    o.column = None
    if x is None:
        o.literal_type = abi.TYPE_NULL
        return o
    o.literal_type, value = x
    if o.literal_type == abi.TYPE_INT:
        o.literal.i32 = int(value)
    elif o.literal_type == abi.TYPE_LONG:
        o.literal.i64 = int(value)
    elif o.literal_type == abi.TYPE_FLOAT:
        o.literal.f32 = float(value)
    else:
        o.literal.f64 = float(value)
    return o


def projection_arithmetic(op, left, right):
    """left <op> right (abi.ARITH_*), operands: columns or literals -> ResultColumn (stays on the device)."""
    lib = abi.load_library()
    handle = C.c_void_p()
    lo, ro = _operand(left), _operand(right)
    abi.check(lib.hy_projection_arithmetic(op, C.byref(lo), C.byref(ro), C.byref(handle)))
    return ResultColumn(handle)


class HostJoinResult:
    """Join result in host memory (numpy views): pairs[k] = (left RowID, right RowID), slice boundaries."""

    def __init__(self, capacity, slice_capacity, radix_bits=None):
        self.left = np.zeros((max(1, capacity), 2), dtype=np.uint32)
        self.right = np.zeros((max(1, capacity), 2), dtype=np.uint32)
        self.slice_offsets = np.zeros(slice_capacity + 2, dtype=np.uint64)
        r = abi.JoinResult()
        r.mem = abi.MEM_HOST
        r.radix_bits = 0xFFFFFFFF if radix_bits is None else radix_bits
        r.left_pos, r.right_pos = self.left.ctypes.data, self.right.ctypes.data
        r.capacity = capacity
        r.slice_offsets = self.slice_offsets.ctypes.data
        r.slice_capacity = slice_capacity
        self.c = r

    @property
    def n_pairs(self):
        return int(self.c.n_pairs)


def join_hash_count(left, right, mode):
    lib = abi.load_library()
    n = C.c_uint64(0)
    abi.check(lib.hy_join_hash_count(left.handle, right.handle, mode, C.byref(n)))
    return int(n.value)


def join_predicates(secondary):
    """[(left DeviceColumn, condition, right DeviceColumn), ...] -> (ctypes array or None, count)"""
    if not secondary:
        return None, 0
    array = (abi.JoinPredicate * len(secondary))()
    for i, (left_column, condition, right_column) in enumerate(secondary):
        array[i].left_column, array[i].right_column, array[i].condition = left_column.handle, right_column.handle, condition
    return array, len(secondary)


def join_hash(left, right, mode, radix_bits=None, secondary=None, capacity=None):
    """hy_join_hash (hy_join_hash_predicates with secondary predicates) with a host-memory result, in ONE call: the result is
    sized for one partner per row of the larger input (every key / foreign-key join fits); a join that multiplies rows answers
    HY_ERR_CAPACITY with what it needs (nothing written) and runs once more with exactly that -- never a counting call first."""
    lib = abi.load_library()
    predicates, n = join_predicates(secondary)
    capacity = max(1, left.rows, right.rows) if capacity is None else capacity
    slice_capacity = max(left.rows, right.rows) // 131070 + max(left.n_chunks, right.n_chunks) + 600
    for attempt in (0, 1):
        result = HostJoinResult(capacity, slice_capacity, radix_bits)
        status = lib.hy_join_hash_predicates(left.handle, right.handle, mode, predicates, n, C.byref(result.c)) if n else \
            lib.hy_join_hash(left.handle, right.handle, mode, C.byref(result.c))
        if status == abi.ERR_CAPACITY and attempt == 0 and (result.n_pairs > capacity or int(result.c.n_slices) > slice_capacity):
            capacity, slice_capacity = max(capacity, result.n_pairs), max(slice_capacity, int(result.c.n_slices))
            continue
        abi.check(status)
        return result


class HostAggregateResult:
    """Aggregate result in host memory.  values of aggregate a: .column(a) (None = NULL)."""
    _NP = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}

    def __init__(self, n_aggregates, group_capacity):
        self.row_ids = np.zeros((max(1, group_capacity), 2), dtype=np.uint32)
        self.raw = [np.zeros(max(1, group_capacity), dtype=np.uint64) for _ in range(n_aggregates)]
        self.nulls = [np.zeros(max(1, group_capacity), dtype=np.uint8) for _ in range(n_aggregates)]
        self.columns = (abi.AggregateColumn * max(1, n_aggregates))()
        for a in range(n_aggregates):
            self.columns[a].values = self.raw[a].ctypes.data
            self.columns[a].is_null = self.nulls[a].ctypes.data
        r = abi.AggregateResult()
        r.mem = abi.MEM_HOST
        r.group_capacity = group_capacity
        r.group_row_ids = self.row_ids.ctypes.data
        r.columns = self.columns
        self.c = r

    @property
    def n_groups(self):
        return int(self.c.n_groups)

    def column(self, a):
        t = self._NP[self.columns[a].data_type]
        n = self.n_groups
        values = np.frombuffer(self.raw[a].tobytes(), dtype=t)[:n]
        return [None if self.nulls[a][i] else values[i].item() for i in range(n)]


def aggregate_hash(groupby_columns, aggregates, group_capacity=None, result=None):
    """aggregates: list of (HY_AGG_*, DeviceColumn or None for COUNT(*)).  result: a HostAggregateResult of an earlier call with the same
    aggregates and capacity, written again (a caller that runs the plan repeatedly keeps its buffers; fresh numpy arrays are first touched --
    one page fault per 4 KiB -- while the result is copied into them)."""
    lib = abi.load_library()
    garr = (C.c_void_p * max(1, len(groupby_columns)))(*[c.handle for c in groupby_columns])
    specs = (abi.AggregateSpec * max(1, len(aggregates)))()
    for i, (function, column) in enumerate(aggregates):
        specs[i].function = function
        specs[i].column = column.handle if column is not None else None
    shape = groupby_columns[0] if groupby_columns else next(c for _, c in aggregates if c is not None)
    if result is None:
        result = HostAggregateResult(len(aggregates), (shape.rows + 1) if group_capacity is None else group_capacity)
    abi.check(lib.hy_aggregate_hash(garr, len(groupby_columns), specs, len(aggregates), C.byref(result.c)))
    return result


def star_join_aggregate(dimensions, groupby, aggregates, group_capacity=4096, result=None):
    """hy_star_join_aggregate: the star join fact x dimensions -> GROUP BY -> aggregates as one call (csrc/plan.hip).
    dimensions: [(key column, filter column or None, predicate or None, fact foreign-key column)]; groupby: [(table, column)] with table 0 =
    the fact table, d + 1 = dimension d; aggregates: [(function, (table, column) or None for COUNT(*), op or None, (table, column) or None)].
    -> (HostAggregateResult, rows of the join result)"""
    lib = abi.load_library()
    dims = (abi.StarDimension * len(dimensions))()
    for d, (key, filter_column, predicate, fact_key) in enumerate(dimensions):
        dims[d].key, dims[d].fact_key = key.handle, fact_key.handle
        dims[d].filter_column = filter_column.handle if filter_column is not None else None
        if predicate is not None:
            dims[d].predicate = predicate
    groups = (abi.StarColumn * max(1, len(groupby)))()
    for g, (table, column) in enumerate(groupby):
        groups[g].table, groups[g].column = table, column.handle
    specs = (abi.StarAggregate * max(1, len(aggregates)))()
    for a, (function, left, op, right) in enumerate(aggregates):
        specs[a].function, specs[a].op = function, abi.STAR_NO_OP if op is None else op
        if left is not None:
            specs[a].left.table, specs[a].left.column = left[0], left[1].handle
        if right is not None:
            specs[a].right.table, specs[a].right.column = right[0], right[1].handle
    if result is None:
        result = HostAggregateResult(len(aggregates), group_capacity)
    joined = C.c_uint64(0)
    abi.check(lib.hy_star_join_aggregate(dims, len(dimensions), groups, len(groupby), specs, len(aggregates), C.byref(result.c), C.byref(joined)))
    return result, int(joined.value)


def expression(tree):
    """An arithmetic expression as hy_expression (postfix).  tree: a column (DeviceColumn), a literal (HY_TYPE_*, value), None (the
    NULL literal) or (abi.ARITH_*, left tree, right tree)."""
    e = abi.Expression()
    nodes = []

    def walk(t):
        n = abi.ExpressionNode()
        if hasattr(t, "handle"):
            n.kind, n.column = abi.EXPR_COLUMN, t.handle
        elif t is None or len(t) == 2:
            literal = _operand(t)
            n.kind, n.literal_type, n.literal = abi.EXPR_LITERAL, literal.literal_type, literal.literal
        else:
            op, left, right = t
            walk(left)
            walk(right)
            n.kind, n.op = abi.EXPR_ARITHMETIC, op
        nodes.append(n)

    walk(tree)
    if len(nodes) > abi.MAX_EXPRESSION_NODES:
        raise ValueError(f"an expression has at most {abi.MAX_EXPRESSION_NODES} nodes")
    e.n_nodes = len(nodes)
    for i, n in enumerate(nodes):
        e.nodes[i] = n
    return e


PAIR_LIST_PERIOD = 2 << 20          # the two output PosLists of a join are written at the same pair index at the same time: when their
PAIR_LIST_OFFSET = 5 << 18          # addresses are congruent modulo 2 MiB the two streams meet in the same memory channels -- pk_emit takes
                                    # 306 - 315 us; 1.25 MiB apart (mod 2 MiB) 274 - 283 us (tools/join_placement.py, profiles/r03_join_placement.txt)


def pair_lists(torch, device, capacity):
    """Device buffers for a join's two output PosLists ([capacity, 2] int32 each), carved out of ONE allocation so that the second starts
    1.25 MiB past a 2 MiB boundary when the first starts on one (callers that hold torch tensors; the C++ adapter and bench.py take
    their lists from the library's pool, hy_result_pool_acquire_pair, which applies the same rule).
    -> (left, right, the allocation: keep it alive)"""
    list_bytes = 8 * max(1, int(capacity))
    arena = torch.empty(2 * list_bytes + 3 * PAIR_LIST_PERIOD, dtype=torch.uint8, device=device)
    first = -arena.data_ptr() % PAIR_LIST_PERIOD
    second = (first + list_bytes + PAIR_LIST_PERIOD - 1) // PAIR_LIST_PERIOD * PAIR_LIST_PERIOD + PAIR_LIST_OFFSET
    rows = max(1, int(capacity))
    left = arena[first:first + list_bytes].view(torch.int32).view(rows, 2)
    right = arena[second:second + list_bytes].view(torch.int32).view(rows, 2)
    return left, right, arena


def validate_filter(mvcc_column, our_tid, snapshot_commit_id, can_use_chunk_shortcut=True):
    """Validate as a filter of hy_scan_project_aggregate: (the table's MvccData as a DeviceColumn, its predicate)."""
    predicate = abi.Predicate()
    predicate.condition = abi.FILTER_VALIDATE
    predicate.value.value_id, predicate.value2.value_id = int(our_tid), int(snapshot_commit_id)
    predicate.column_is_nullable = 1 if can_use_chunk_shortcut else 0
    return mvcc_column, predicate


def scan_project_aggregate(filters, groupby_columns, aggregates, group_capacity=None):
    """hy_scan_project_aggregate: TableScan(s) -> Projection -> AggregateHash of one data table in one pass.
    filters: [(DeviceColumn, predicate)], ANDed in order; aggregates: [(HY_AGG_*, expression tree or None for COUNT(*))]."""
    lib = abi.load_library()
    farr = (abi.Filter * max(1, len(filters)))()
    for i, (column, predicate) in enumerate(filters):
        farr[i].column = column.handle
        farr[i].predicate = predicate          # (a copy of the struct; `filters` keeps the arrays it points to alive)
    garr = (C.c_void_p * max(1, len(groupby_columns)))(*[c.handle for c in groupby_columns])
    expressions = [expression(tree) if tree is not None or function != abi.AGG_COUNT else None for function, tree in aggregates]
    specs = (abi.FusedAggregate * max(1, len(aggregates)))()
    for i, (function, _) in enumerate(aggregates):
        specs[i].function = function
        if expressions[i] is not None:
            specs[i].input = C.pointer(expressions[i])

    def columns_of(tree):
        if hasattr(tree, "handle"):
            yield tree
        elif tree is not None and len(tree) == 3:
            yield from columns_of(tree[1])
            yield from columns_of(tree[2])

    table = [c for c, _ in filters] + list(groupby_columns) + [c for _, tree in aggregates for c in columns_of(tree)]
    if not table:
        raise ValueError("hy_scan_project_aggregate needs at least one column")
    result = HostAggregateResult(len(aggregates), (table[0].rows + 1) if group_capacity is None else group_capacity)
    abi.check(lib.hy_scan_project_aggregate(farr, len(filters), garr, len(groupby_columns), specs, len(aggregates), C.byref(result.c)))
    return result
