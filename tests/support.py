"""Test support: the CPU oracle binding (oracle/liboracle.so -- TEST INFRASTRUCTURE, never imported by the product),
the reference's `.tbl` text-table format, and helpers to run one predicate through oracle and device."""
import ctypes as C
import os
import subprocess

import numpy as np

from hyrise_amd import abi, storage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "tbl")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OracleColumn(C.Structure):
    _fields_ = [("segments", C.POINTER(abi.Segment)), ("n_chunks", C.c_uint32)]


_oracle = None


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(ORACLE_PATH)
    lib.hyo_encode_dictionary.restype = C.c_uint32
    lib.hyo_encode_dictionary.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_uint32)]
    lib.hyo_encode_frame_of_reference.restype = C.c_uint32
    lib.hyo_encode_frame_of_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.c_uint32)]
    lib.hyo_pack_nulls.restype = None
    lib.hyo_pack_nulls.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hyo_scan_chunk.restype = C.c_int64
    lib.hyo_scan_chunk.argtypes = [C.POINTER(OracleColumn), C.c_uint32, C.POINTER(abi.Predicate), C.c_void_p, C.c_void_p]
    lib.hyo_scan_chunk_columns.restype = C.c_int64
    lib.hyo_scan_chunk_columns.argtypes = [C.POINTER(OracleColumn), C.POINTER(OracleColumn), C.c_uint32, C.c_uint32,
                                           C.c_void_p]
    lib.hyo_table_scan.restype = C.c_int32
    lib.hyo_table_scan.argtypes = [C.POINTER(OracleColumn), C.POINTER(abi.Predicate), C.POINTER(abi.ScanResult), C.c_int]
    lib.hyo_arithmetic.restype = C.c_uint32
    lib.hyo_arithmetic.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                   C.c_void_p, C.c_void_p]
    lib.hyo_expression_common_type.restype = C.c_uint32
    lib.hyo_expression_common_type.argtypes = [C.c_uint32, C.c_uint32]
    lib.hyo_validate.restype = C.c_int32
    lib.hyo_validate.argtypes = [C.POINTER(OracleColumn), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(abi.ScanResult)]
    lib.hyo_table_scan_columns.restype = C.c_int32
    lib.hyo_table_scan_columns.argtypes = [C.POINTER(OracleColumn), C.POINTER(OracleColumn), C.c_uint32,
                                           C.POINTER(abi.ScanResult), C.c_int]
    _oracle = lib
    return lib


class OracleCol:
    """hyo_column over a HostColumn (host pointers)."""

    def __init__(self, host_column):
        from hyrise_amd.storage import expand_run_length
        host_column = expand_run_length(host_column)   # the oracle reads Value / Dictionary / FrameOfReference segments
        self.host = host_column
        self._ref_cols = {}

        def resolve(ref_host):
            col = self._ref_cols.get(id(ref_host))
            if col is None:
                col = OracleCol(ref_host)
                self._ref_cols[id(ref_host)] = col
            return C.addressof(col.c)

        self._descriptors = host_column.descriptors(resolve)
        self.c = OracleColumn(self._descriptors, host_column.n_chunks)


def oracle_scan(host_column, predicate, flags=0, threads=1):
    """hyo_table_scan -> same result layout as the ABI (host numpy arrays)."""
    from hyrise_amd.operators import HostScanResult
    col = OracleCol(host_column)
    result = HostScanResult(host_column.n_chunks, host_column.rows, flags)
    status = oracle().hyo_table_scan(C.byref(col.c), C.byref(predicate), C.byref(result.c), threads)
    assert status == 0, f"oracle scan failed with {status}"
    return result


def oracle_validate(host_column, our_tid, snapshot_commit_id, can_use_chunk_shortcut=True, flags=0):
    from hyrise_amd.operators import HostScanResult
    col = OracleCol(host_column)
    result = HostScanResult(host_column.n_chunks, host_column.rows, flags)
    status = oracle().hyo_validate(C.byref(col.c), our_tid, snapshot_commit_id, 1 if can_use_chunk_shortcut else 0, C.byref(result.c))
    assert status == 0, f"oracle validate failed with {status}"
    return result


def oracle_arithmetic(op, left, right, n=None):
    """left / right: (numpy values, bool nulls or None) for a column, (HY_TYPE_*, value) for a literal, None for a NULL
    literal.  Returns (values, nulls) as the reference's ExpressionEvaluator produces them."""
    def operand(x):
        if x is None:
            return abi.TYPE_NULL, np.zeros(1, dtype=np.int32), np.ones(1, dtype=np.uint8), 0
        if isinstance(x[0], np.ndarray):
            values = np.ascontiguousarray(x[0])
            nulls = np.ascontiguousarray(x[1], dtype=np.uint8) if x[1] is not None else None
            return storage.TYPE_OF_NP[values.dtype], values, nulls, 1
        data_type, value = x
        return data_type, np.array([value], dtype=storage.NP_TYPES[data_type]), None, 0
    lt, lv, ln, ls = operand(left)
    rt, rv, rn, rs = operand(right)
    if n is None:
        n = len(lv) if ls else len(rv)
    result_type = oracle().hyo_expression_common_type(lt, rt)
    out = np.zeros(n, dtype=storage.NP_TYPES[result_type])
    out_nulls = np.zeros(n, dtype=np.uint8)
    got = oracle().hyo_arithmetic(op, lt, lv.ctypes.data, ln.ctypes.data if ln is not None else None, ls, rt, rv.ctypes.data,
                                  rn.ctypes.data if rn is not None else None, rs, n, out.ctypes.data, out_nulls.ctypes.data)
    assert got == result_type
    return out, out_nulls.astype(bool)


def oracle_scan_columns(left, right, condition, threads=1):
    from hyrise_amd.operators import HostScanResult
    lcol, rcol = OracleCol(left), OracleCol(right)
    result = HostScanResult(left.n_chunks, left.rows)
    status = oracle().hyo_table_scan_columns(C.byref(lcol.c), C.byref(rcol.c), condition, C.byref(result.c), threads)
    assert status == 0, f"oracle scan failed with {status}"
    return result


# ---- .tbl text tables (src/lib/utils/load_table.cpp:22-96) -----------------------------------------------------------
TBL_TYPES = {"int": abi.TYPE_INT, "long": abi.TYPE_LONG, "float": abi.TYPE_FLOAT, "double": abi.TYPE_DOUBLE,
             "string": abi.TYPE_STRING}


class TblTable:
    def __init__(self, names, types, nullable, columns, nulls):
        self.names, self.types, self.nullable, self.columns, self.nulls = names, types, nullable, columns, nulls

    @property
    def rows(self):
        return len(self.columns[0]) if self.columns else 0

    def column(self, name_or_index):
        i = self.names.index(name_or_index) if isinstance(name_or_index, str) else name_or_index
        return self.columns[i], (self.nulls[i] if self.nullable[i] else None)


def load_tbl(path):
    """Line 1: column names, line 2: types ('int', 'float_null', ...), then '|'-separated rows; 'null' in a nullable
    column is NULL."""
    if not os.path.isabs(path):
        path = os.path.join(GOLDEN, path)
    with open(path) as fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    names = lines[0].split("|")
    type_specs = lines[1].split("|")
    types, nullable = [], []
    for spec in type_specs:
        parts = spec.split("_")
        types.append(TBL_TYPES[parts[0]])
        nullable.append(len(parts) > 1 and parts[1] == "null")
    raw = [line.split("|") for line in lines[2:]]
    columns, nulls = [], []
    for c, t in enumerate(types):
        cells = [row[c] for row in raw]
        is_null = np.array([nullable[c] and cell == "null" for cell in cells], dtype=bool)
        if t == abi.TYPE_STRING:
            values = np.array([("" if n else cell) for cell, n in zip(cells, is_null)], dtype=object)
        else:
            np_type = storage.NP_TYPES[t]
            if t in (abi.TYPE_INT, abi.TYPE_LONG):
                values = np.array([0 if n else int(cell) for cell, n in zip(cells, is_null)], dtype=np_type)
            else:
                values = np.array([0 if n else float(cell) for cell, n in zip(cells, is_null)], dtype=np_type)
        columns.append(values)
        nulls.append(is_null)
    return TblTable(names, types, nullable, columns, nulls)


def decode_rows(host_column, rows):
    """Values (None for NULL) at (chunk_id, chunk_offset) RowIDs of a DATA HostColumn -- test-side decoding."""
    out = []
    for chunk_id, offset in rows:
        seg = host_column.segments[int(chunk_id)]
        offset = int(offset)
        if seg.encoding == abi.ENC_DICTIONARY:
            vid = int(seg.data[offset])
            out.append(None if vid == seg.aux_size else seg.aux[vid].item())
        elif seg.encoding == abi.ENC_FRAME_OF_REFERENCE:
            is_null = seg.nulls is not None and (int(seg.nulls[offset // 64]) >> (offset % 64)) & 1
            out.append(None if is_null else int(seg.data[offset]) + int(seg.aux[offset // abi.FOR_BLOCK_SIZE]))
        else:
            is_null = seg.nulls is not None and (int(seg.nulls[offset // 64]) >> (offset % 64)) & 1
            out.append(None if is_null else seg.data[offset].item())
    return out


def assert_scan_equal(device_result, oracle_result, context=""):
    """Bit-exact comparison of two scan results (offsets, counts, states, RowIDs)."""
    n = oracle_result.n_chunks
    np.testing.assert_array_equal(device_result.counts[:n], oracle_result.counts[:n], err_msg=f"counts {context}")
    np.testing.assert_array_equal(device_result.chunk_state[:n], oracle_result.chunk_state[:n], err_msg=f"states {context}")
    np.testing.assert_array_equal(device_result.offsets, oracle_result.offsets, err_msg=f"offsets {context}")
    total = oracle_result.total
    assert device_result.matches[:total].tobytes() == oracle_result.matches[:total].tobytes(), f"PosLists differ {context}"


def gpu_available():
    try:
        lib = abi.load_library()
    except (ImportError, OSError):
        return False
    count = C.c_int32(0)
    return lib.hy_device_count(C.byref(count)) == 0 and count.value > 0


def build_column(values, nulls, chunk_size, encodings, nullable=None):
    """Column with per-chunk encodings (int or list; chunks past the list stay unencoded, like the reference's
    partly-compressed test tables)."""
    values = np.ascontiguousarray(values)
    n = len(values)
    nullable = (nulls is not None) if nullable is None else nullable
    segments = []
    for c, begin in enumerate(range(0, n, chunk_size)):
        end = min(n, begin + chunk_size)
        enc = encodings if isinstance(encodings, int) else (encodings[c] if c < len(encodings) else abi.ENC_UNENCODED)
        if enc == abi.ENC_FRAME_OF_REFERENCE and values.dtype != np.int32:
            enc = abi.ENC_UNENCODED  # encoding_supports_data_type(): fall back like load_and_encode_table
        chunk_nulls = None
        if nulls is not None:
            chunk_nulls = np.asarray(nulls[begin:end], dtype=bool)
        elif nullable and enc == abi.ENC_UNENCODED:
            chunk_nulls = np.zeros(end - begin, dtype=bool)
        segments.append(storage.encode_segment(values[begin:end], chunk_nulls, enc))
    return storage.HostColumn(segments, storage.TYPE_OF_NP[values.dtype])


def brute_force_scan(values, nulls, condition, value=None, value2=None):
    """Independent numpy evaluation of a predicate (SQL semantics: NULL never matches a comparison)."""
    valid = ~nulls if nulls is not None else np.ones(len(values), dtype=bool)
    v = values
    if condition == abi.PRED_IS_NULL:
        return ~valid
    if condition == abi.PRED_IS_NOT_NULL:
        return valid
    t = values.dtype.type
    a = t(value)
    ops = {abi.PRED_EQUALS: v == a, abi.PRED_NOT_EQUALS: v != a, abi.PRED_LESS_THAN: v < a,
           abi.PRED_LESS_THAN_EQUALS: v <= a, abi.PRED_GREATER_THAN: v > a, abi.PRED_GREATER_THAN_EQUALS: v >= a}
    if condition in ops:
        return ops[condition] & valid
    b = t(value2)
    lower = (v >= a) if condition in (abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_BETWEEN_UPPER_EXCLUSIVE) else (v > a)
    upper = (v <= b) if condition in (abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_BETWEEN_LOWER_EXCLUSIVE) else (v < b)
    return lower & upper & valid


def expected_result_from_mask(mask, chunk_size):
    """(chunk_id, chunk_offset) pairs, ascending, for a boolean row mask of a data table."""
    rows = np.nonzero(mask)[0]
    return np.stack([rows // chunk_size, rows % chunk_size], axis=1).astype(np.uint32)


def result_rows(result):
    """All matching (chunk_id, chunk_offset) RowIDs of a scan result, expanding ALL_MATCH chunks (for which the ABI
    writes no RowIDs: the adapter emits an EntireChunkPosList, table_scan.cpp:201-205)."""
    rows = []
    for c in range(result.n_chunks):
        if result.chunk_state[c] == abi.CHUNK_ALL_MATCH and result.offsets[c + 1] == result.offsets[c]:
            rows.extend((c, i) for i in range(int(result.counts[c])))
        else:
            rows.extend(map(tuple, result.pos_list(c).tolist()))
    return rows


class DeviceArray:
    """A device buffer through the C ABI (hy_device_malloc / hy_memcpy_d2h): the tests need no torch."""

    def __init__(self, lib, shape, dtype):
        self.lib, self.shape, self.dtype = lib, shape, np.dtype(dtype)
        self.nbytes = int(np.prod(shape)) * self.dtype.itemsize
        pointer = C.c_void_p()
        abi.check(lib.hy_device_malloc(C.byref(pointer), max(self.nbytes, 256)))
        self.pointer = pointer.value

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        abi.check(self.lib.hy_memcpy_d2h(out.ctypes.data, self.pointer, self.nbytes))
        return out

    def __del__(self):
        if getattr(self, "pointer", None):
            self.lib.hy_device_free(self.pointer)
            self.pointer = None


# ---- JoinHash -------------------------------------------------------------------------------------------------------
class HostJoinResult:
    def __init__(self, capacity, slice_capacity, radix_bits=None, mem=abi.MEM_HOST):
        self.left = np.zeros((max(1, capacity), 2), dtype=np.uint32)
        self.right = np.zeros((max(1, capacity), 2), dtype=np.uint32)
        self.slice_offsets = np.zeros(slice_capacity + 2, dtype=np.uint64)
        r = abi.JoinResult()
        r.mem = mem
        r.radix_bits = 0xFFFFFFFF if radix_bits is None else radix_bits
        r.left_pos, r.right_pos = self.left.ctypes.data, self.right.ctypes.data
        r.capacity = capacity
        r.slice_offsets = self.slice_offsets.ctypes.data
        r.slice_capacity = slice_capacity
        self.c = r

    @property
    def n_pairs(self):
        return int(self.c.n_pairs)

    def pairs(self):
        n = self.n_pairs
        return self.left[:n], self.right[:n]


def _bind_join(lib):
    lib.hyo_join_hash.restype = C.c_int32
    lib.hyo_join_hash.argtypes = [C.POINTER(OracleColumn), C.POINTER(OracleColumn), C.c_uint32, C.POINTER(abi.JoinResult), C.c_int]
    lib.hyo_join_hash_predicates.restype = C.c_int32
    lib.hyo_join_hash_predicates.argtypes = [C.POINTER(OracleColumn), C.POINTER(OracleColumn), C.c_uint32, C.POINTER(abi.JoinPredicate), C.c_uint32,
                                             C.POINTER(abi.JoinResult), C.c_int]
    lib.hyo_calculate_radix_bits.restype = C.c_uint32
    lib.hyo_calculate_radix_bits.argtypes = [C.c_uint64, C.c_uint64]
    lib.hyo_join_materialize.restype = C.c_uint64
    lib.hyo_join_materialize.argtypes = [C.POINTER(OracleColumn), C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def oracle_join(left, right, mode, radix_bits=None, threads=1, capacity=None, secondary=None):
    """secondary: [(left HostColumn, condition, right HostColumn), ...]"""
    lib = oracle()
    _bind_join(lib)
    lcol, rcol = OracleCol(left), OracleCol(right)
    extra = [(OracleCol(l), c, OracleCol(r)) for l, c, r in (secondary or [])]
    predicates = (abi.JoinPredicate * max(1, len(extra)))()
    for i, (l, c, r) in enumerate(extra):
        predicates[i].left_column, predicates[i].right_column, predicates[i].condition = C.addressof(l.c), C.addressof(r.c), c
    if capacity is None:
        capacity = max(left.rows, right.rows, 1) * 4 + 1024
    slice_capacity = (max(left.rows, right.rows) // 131070) + max(len(left.segments), len(right.segments)) + 600   # (without radix partitioning: a PosList per probe chunk)
    for attempt in range(6):
        result = HostJoinResult(capacity, slice_capacity, radix_bits)
        status = lib.hyo_join_hash_predicates(C.byref(lcol.c), C.byref(rcol.c), mode, predicates, len(extra), C.byref(result.c), threads)
        if status == abi.ERR_CAPACITY:   # (the oracle does not say which: first more PosLists -- a partial one per radix partition --, then more pairs)
            if attempt == 0:
                slice_capacity += 1 << 17
            else:
                capacity *= 8
            continue
        assert status == 0, f"oracle join failed with {status}"
        return result
    raise AssertionError("oracle join: the result does not fit")


def column_values(host_column):
    """(values list with None for NULL) of a HostColumn, row order, decoding data or reference segments."""
    out = []
    for c, seg in enumerate(host_column.segments):
        if seg.encoding == abi.ENC_REFERENCE:
            if seg.data is None:
                rows = [(seg.ref_chunk_id, i) for i in range(seg.size)]
            else:
                rows = [tuple(r) for r in seg.data.tolist()]
            for r in rows:
                out.append(None if r[1] == 0xFFFFFFFF else decode_rows(seg.ref, [r])[0])
        else:
            out.extend(decode_rows(host_column, [(c, i) for i in range(seg.size)]))
    return out


def row_ids_of(host_column):
    rows = []
    for c, seg in enumerate(host_column.segments):
        rows.extend((c, i) for i in range(seg.size))
    return rows


_NP_OF_TYPE = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}


def cxx_compare(condition, x, x_type, y, y_type):
    """x <condition> y the way C++ compares an x_type with a y_type value (usual arithmetic conversions: the comparator
    functors of type_comparison.hpp are generic lambdas) -- int64 against float compares as float."""
    if abi.TYPE_DOUBLE in (x_type, y_type):
        common = np.float64
    elif abi.TYPE_FLOAT in (x_type, y_type):
        common = np.float32
    else:
        common = np.int64
    a, b = common(_NP_OF_TYPE[x_type](x)), common(_NP_OF_TYPE[y_type](y))
    return {abi.PRED_EQUALS: a == b, abi.PRED_NOT_EQUALS: a != b, abi.PRED_LESS_THAN: a < b, abi.PRED_LESS_THAN_EQUALS: a <= b,
            abi.PRED_GREATER_THAN: a > b, abi.PRED_GREATER_THAN_EQUALS: a >= b}[condition]


def _verification_join_general(left, right, mode, secondary):
    """The nested loops themselves, with secondary predicates (join_verification.cpp:60-184): a pair matches if the keys
    are equal and every secondary predicate holds (NULL on either side: it does not)."""
    lv, rv = column_values(left), column_values(right)
    lrows, rrows = row_ids_of(left), row_ids_of(right)
    extra = [(column_values(l), l.data_type, c, column_values(r), r.data_type) for l, c, r in secondary]

    def matches(i, j):   # the keys compare like C++ compares the two column types (= after JoinHash's cast to its HashedType)
        if lv[i] is None or rv[j] is None or not cxx_compare(abi.PRED_EQUALS, lv[i], left.data_type, rv[j], right.data_type):
            return False
        for xs, xt, c, ys, yt in extra:
            if xs[i] is None or ys[j] is None or not cxx_compare(c, xs[i], xt, ys[j], yt):
                return False
        return True

    out = []
    if mode == abi.JOIN_INNER:
        out = [(lrows[i], rrows[j]) for i in range(len(lv)) for j in range(len(rv)) if matches(i, j)]
    elif mode == abi.JOIN_LEFT:
        for i in range(len(lv)):
            found = [(lrows[i], rrows[j]) for j in range(len(rv)) if matches(i, j)]
            out += found or [(lrows[i], None)]
    elif mode == abi.JOIN_RIGHT:
        for j in range(len(rv)):
            found = [(lrows[i], rrows[j]) for i in range(len(lv)) if matches(i, j)]
            out += found or [(None, rrows[j])]
    elif mode == abi.JOIN_SEMI:
        out = [(lrows[i], None) for i in range(len(lv)) if any(matches(i, j) for j in range(len(rv)))]
    elif mode == abi.JOIN_ANTI_NULL_AS_FALSE:
        out = [(lrows[i], None) for i in range(len(lv)) if not any(matches(i, j) for j in range(len(rv)))]
    elif not secondary:   # AntiNullAsTrue: a NULL on either side counts as a match
        def maybe(i, j):
            return lv[i] is None or rv[j] is None or matches(i, j)
        out = [(lrows[i], None) for i in range(len(lv)) if not any(maybe(i, j) for j in range(len(rv)))]
    else:
        raise ValueError("AntiNullAsTrue has no secondary predicates in JoinHash")
    return sorted(out, key=lambda p: (p[0] is None, p[0] or (0, 0), p[1] is None, p[1] or (0, 0)))


def verification_join(left, right, mode, secondary=None):
    """Nested-loop reference of the join semantics (JoinVerification, operators/join_verification.cpp:60-184), as a
    sorted multiset of ((left chunk, left offset) | None, (right chunk, right offset) | None)."""
    floating = (abi.TYPE_FLOAT, abi.TYPE_DOUBLE)
    if secondary or left.data_type in floating or right.data_type in floating:
        return _verification_join_general(left, right, mode, secondary or [])
    lv, rv = column_values(left), column_values(right)
    lrows, rrows = row_ids_of(left), row_ids_of(right)
    out = []
    if mode == abi.JOIN_INNER:
        index = {}
        for j, v in enumerate(rv):
            if v is not None:
                index.setdefault(v, []).append(j)
        for i, v in enumerate(lv):
            for j in index.get(v, []) if v is not None else []:
                out.append((lrows[i], rrows[j]))
    elif mode in (abi.JOIN_LEFT, abi.JOIN_RIGHT):
        outer_v, outer_rows, inner_v, inner_rows = (lv, lrows, rv, rrows) if mode == abi.JOIN_LEFT else (rv, rrows, lv, lrows)
        index = {}
        for j, v in enumerate(inner_v):
            if v is not None:
                index.setdefault(v, []).append(j)
        for i, v in enumerate(outer_v):
            matches = index.get(v, []) if v is not None else []
            for j in matches:
                out.append((outer_rows[i], inner_rows[j]) if mode == abi.JOIN_LEFT else (inner_rows[j], outer_rows[i]))
            if not matches:
                out.append((outer_rows[i], None) if mode == abi.JOIN_LEFT else (None, outer_rows[i]))
    else:
        rset = set(v for v in rv if v is not None)
        right_has_null = any(v is None for v in rv)
        for i, v in enumerate(lv):
            if mode == abi.JOIN_SEMI:
                keep = v is not None and v in rset
            elif mode == abi.JOIN_ANTI_NULL_AS_FALSE:
                keep = v is None or v not in rset
            else:  # AntiNullAsTrue: NULL comparisons count as matches
                if len(rv) == 0:
                    keep = True
                elif v is None or right_has_null:
                    keep = False
                else:
                    keep = v not in rset
            if keep:
                out.append((lrows[i], None))
    return sorted(out, key=lambda p: (p[0] is None, p[0] or (0, 0), p[1] is None, p[1] or (0, 0)))


def join_result_multiset(result, mode):
    n = result.n_pairs
    semi = mode in (abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_ANTI_NULL_AS_FALSE)
    out = []
    for k in range(n):
        l = tuple(int(x) for x in result.left[k])
        left_id = None if l[1] == 0xFFFFFFFF else l
        if semi:
            out.append((left_id, None))
        else:
            r = tuple(int(x) for x in result.right[k])
            out.append((left_id, None if r[1] == 0xFFFFFFFF else r))
    return sorted(out, key=lambda p: (p[0] is None, p[0] or (0, 0), p[1] is None, p[1] or (0, 0)))


# ---- AggregateHash --------------------------------------------------------------------------------------------------
AGG_BY_NAME = {"Min": abi.AGG_MIN, "Max": abi.AGG_MAX, "Sum": abi.AGG_SUM, "Avg": abi.AGG_AVG, "Count": abi.AGG_COUNT,
               "CountDistinct": abi.AGG_COUNT_DISTINCT, "StandardDeviationSample": abi.AGG_STDDEV_SAMP, "Any": abi.AGG_ANY}
_RESULT_NP = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}


class HostAggregateResult:
    def __init__(self, n_aggregates, group_capacity, mem=abi.MEM_HOST):
        self.row_ids = np.zeros((max(1, group_capacity), 2), dtype=np.uint32)
        self.raw = [np.zeros(max(1, group_capacity), dtype=np.uint64) for _ in range(n_aggregates)]
        self.nulls = [np.zeros(max(1, group_capacity), dtype=np.uint8) for _ in range(n_aggregates)]
        self.columns = (abi.AggregateColumn * max(1, n_aggregates))()
        for a in range(n_aggregates):
            self.columns[a].values = self.raw[a].ctypes.data
            self.columns[a].is_null = self.nulls[a].ctypes.data
        r = abi.AggregateResult()
        r.mem = mem
        r.group_capacity = group_capacity
        r.group_row_ids = self.row_ids.ctypes.data
        r.columns = self.columns
        self.c = r
        self.n_aggregates = n_aggregates

    @property
    def n_groups(self):
        return int(self.c.n_groups)

    def column(self, a):
        """values (python list, None for NULL) of aggregate a."""
        t = _RESULT_NP[self.columns[a].data_type]
        n = self.n_groups
        values = self.raw[a].view(np.uint8)[: 8 * len(self.raw[a])].view(t)[:n] if t in (np.int64, np.float64) else \
            np.frombuffer(self.raw[a].tobytes(), dtype=t)[:n]
        return [None if self.nulls[a][i] else values[i].item() for i in range(n)]


def oracle_aggregate(groupby_columns, aggregates, group_capacity=None):
    """aggregates: list of (function, HostColumn or None)."""
    lib = oracle()
    lib.hyo_aggregate_hash.restype = C.c_int32
    lib.hyo_aggregate_hash.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_uint32,
                                       C.POINTER(abi.AggregateResult)]
    gcols = [OracleCol(c) for c in groupby_columns]
    acols = [OracleCol(c) if c is not None else None for _, c in aggregates]
    garr = (C.c_void_p * max(1, len(gcols)))(*[C.addressof(c.c) for c in gcols])
    aarr = (C.c_void_p * max(1, len(acols)))(*[C.addressof(c.c) if c is not None else None for c in acols])
    farr = (C.c_uint32 * max(1, len(aggregates)))(*[f for f, _ in aggregates])
    rows = (groupby_columns[0] if groupby_columns else next(c for _, c in aggregates if c is not None)).rows
    result = HostAggregateResult(len(aggregates), (rows + 1) if group_capacity is None else group_capacity)
    status = lib.hyo_aggregate_hash(garr, len(gcols), farr, aarr, len(aggregates), C.byref(result.c))
    assert status == 0, f"oracle aggregate failed with {status}"
    result._keep = (gcols, acols)
    return result


# ---- aggregate fixtures with string columns ------------------------------------------------------------------------------
class AggregateCase:
    """The columns of one aggregate_test.cpp case as the device / oracle take them.  String columns travel as int64
    stand-ins (hyrise_amd/string_keys.py): AggregateKeyEntry names where they are GROUP BY columns, order-preserving
    ranks where they are aggregate arguments; `output_rows` maps the result back to the table the reference expects."""

    def __init__(self, case):
        from hyrise_amd import string_keys
        self.case = case
        table = load_tbl(case["input"])
        self.table = table
        encoding = abi.ENC_DICTIONARY if case["encoded"] else abi.ENC_UNENCODED
        chunk = case["chunk_size"]
        self.plain = {}

        def numeric(i):
            if i not in self.plain:
                self.plain[i] = build_column(table.columns[i], table.nulls[i] if table.nullable[i] else None, chunk, encoding)
            return self.plain[i]

        def string_column(i, keys):
            values, nulls = table.columns[i], (table.nulls[i] if table.nullable[i] else None)
            if case["encoded"]:
                segments, dictionaries = string_keys.encode_string_column(values, nulls, chunk)
                return keys.dictionary_column(segments, dictionaries)
            return keys.value_column(values, nulls, chunk)

        self.groupby, self.group_strings = [], []
        for g in case["groupby"]:
            if table.types[g] == abi.TYPE_STRING:
                self.groupby.append(string_column(g, string_keys.AggregateKeyNames()))
            else:
                self.groupby.append(numeric(g))
        self.aggregates, self.ranks = [], []
        for c, f in case["aggregates"]:
            ranks = None
            if c is None:
                column = None
            elif table.types[c] == abi.TYPE_STRING:
                ranks = string_keys.StringRanks(table.columns[c], table.nulls[c] if table.nullable[c] else None)
                column = string_column(c, ranks)
            else:
                column = numeric(c)
            self.aggregates.append((AGG_BY_NAME[f], column))
            self.ranks.append(ranks if AGG_BY_NAME[f] in (abi.AGG_MIN, abi.AGG_MAX, abi.AGG_ANY) else None)

    @property
    def runnable(self):
        return bool(self.groupby) or any(c is not None for _, c in self.aggregates)

    def output_rows(self, result):
        """GROUP BY columns (the values of the representative rows, strings as strings), then one cell per aggregate."""
        table, case = self.table, self.case
        flat_index, offset = {}, 0
        shape = (self.groupby[0] if self.groupby else next(c for _, c in self.aggregates if c is not None))
        for chunk, seg in enumerate(shape.segments):
            for i in range(seg.size):
                flat_index[(chunk, i)] = offset + i
            offset += seg.size
        rows = []
        for g in range(result.n_groups):
            rid = tuple(int(x) for x in result.row_ids[g])
            row = []
            for column_id in case["groupby"]:
                is_null = table.nullable[column_id] and table.nulls[column_id][flat_index[rid]]
                value = table.columns[column_id][flat_index[rid]]
                row.append(None if is_null else (value if table.types[column_id] == abi.TYPE_STRING else value.item()))
            rows.append(row)
        for a in range(len(self.aggregates)):
            cells = result.column(a)
            if self.ranks[a] is not None:
                cells = self.ranks[a].strings_of(cells)
            for g, v in enumerate(cells):
                rows[g].append(v)
        return rows


# ---- TableScan(s) -> Projection -> AggregateHash, operator by operator, on the CPU oracle ----------------------------------------
def oracle_chain(filters, groupby, aggregates):
    """What hy_scan_project_aggregate fuses, run the reference's way: every scan reads the PosLists of the one before
    (table_scan.cpp:158-196), the expressions are materialised node by node over the survivors (oracle_arithmetic), the
    aggregate runs on that reference table.  filters: [(HostColumn, predicate)]; aggregates: [(function, tree or None)], tree =
    HostColumn | (HY_TYPE_*, value) | None | (ARITH_*, tree, tree).
    -> (aggregate result, RowIDs of the data table behind the rows of the aggregate's input table)"""
    from oracle_executor import OracleExecutor
    ex = OracleExecutor()
    lists = None
    for column, predicate in filters:
        lists = ex.scan_chunked(column if lists is None else ex.reference_column_chunked(column, lists), predicate)

    def view(column):
        return column if lists is None else ex.reference_column_chunked(column, lists)

    table = [c for c, _ in filters] + list(groupby)

    def columns_of(tree):
        if hasattr(tree, "segments"):
            yield tree
        elif tree is not None and len(tree) == 3:
            yield from columns_of(tree[1])
            yield from columns_of(tree[2])

    for _, tree in aggregates:
        table.extend(columns_of(tree))
    shape = view(table[0])
    sizes = [s.size for s in shape.segments]
    rows = int(sum(sizes))

    def evaluate(tree):
        if hasattr(tree, "segments"):
            cells = column_values(view(tree))
            nulls = np.array([c is None for c in cells], dtype=bool)
            return np.array([0 if c is None else c for c in cells], dtype=_NP_OF_TYPE[tree.data_type]), (nulls if nulls.any() else None)
        if tree is None or len(tree) == 2:
            return tree
        op, left, right = tree
        values, nulls = oracle_arithmetic(op, evaluate(left), evaluate(right), n=rows)
        return values, (nulls if nulls.any() else None)

    def materialise(tree):
        result = evaluate(tree)
        assert result is not None and isinstance(result[0], np.ndarray), "an aggregate of a literal is not a case of these tests"
        values, nulls = result
        segments, begin = [], 0
        for size in sizes:
            segments.append(storage.encode_segment(values[begin:begin + size], None if nulls is None else nulls[begin:begin + size], abi.ENC_UNENCODED))
            begin += size
        return storage.HostColumn(segments, storage.TYPE_OF_NP[values.dtype])

    inputs = [(function, materialise(tree) if tree is not None else None) for function, tree in aggregates]
    if not groupby and all(c is None for _, c in inputs):   # a lone COUNT(*): the oracle wants one column for the table's shape
        result = oracle_aggregate([], inputs + [(abi.AGG_COUNT, shape)])
    else:
        result = oracle_aggregate([view(c) for c in groupby], inputs)
    if lists is None:
        base_rows = np.array(row_ids_of(shape), dtype=np.uint32).reshape(-1, 2)
    else:
        kept = [rows_ for rows_ in lists.lists if len(rows_)]
        base_rows = np.concatenate(kept).astype(np.uint32).reshape(-1, 2) if kept else np.zeros((0, 2), dtype=np.uint32)
    return result, base_rows, sizes
