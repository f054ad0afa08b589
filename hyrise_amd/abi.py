"""ctypes mirror of include/hyrise_amd.h -- the C ABI of the MI355X hot path.

Python is plumbing only (tests, bench, multi-GPU launch); the product is libhyrise_amd.so.  Loading fails loudly when
the HIP library has not been built: there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhyrise_amd.so")

# enums ---------------------------------------------------------------------------------------------------------------
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_CAPACITY = 0, 1, 2, 3, 4
TYPE_NULL, TYPE_INT, TYPE_LONG, TYPE_FLOAT, TYPE_DOUBLE, TYPE_STRING = range(6)
(PRED_EQUALS, PRED_NOT_EQUALS, PRED_LESS_THAN, PRED_LESS_THAN_EQUALS, PRED_GREATER_THAN, PRED_GREATER_THAN_EQUALS,
 PRED_BETWEEN_INCLUSIVE, PRED_BETWEEN_LOWER_EXCLUSIVE, PRED_BETWEEN_UPPER_EXCLUSIVE, PRED_BETWEEN_EXCLUSIVE,
 PRED_IN, PRED_NOT_IN, PRED_LIKE, PRED_NOT_LIKE, PRED_LIKE_INSENSITIVE, PRED_NOT_LIKE_INSENSITIVE,
 PRED_IS_NULL, PRED_IS_NOT_NULL) = range(18)
(JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL_OUTER, JOIN_CROSS, JOIN_SEMI, JOIN_ANTI_NULL_AS_TRUE,
 JOIN_ANTI_NULL_AS_FALSE) = range(8)
AGG_MIN, AGG_MAX, AGG_SUM, AGG_AVG, AGG_COUNT, AGG_COUNT_DISTINCT, AGG_STDDEV_SAMP, AGG_ANY = range(8)
ENC_UNENCODED, ENC_DICTIONARY, ENC_FRAME_OF_REFERENCE, ENC_REFERENCE, ENC_MVCC, ENC_RUN_LENGTH, ENC_LZ4 = range(7)
SORT_NONE, SORT_ASCENDING_NULLS_FIRST, SORT_DESCENDING_NULLS_FIRST, SORT_ASCENDING_NULLS_LAST, SORT_DESCENDING_NULLS_LAST = range(5)   # hyrise::SortMode + 1
MEM_HOST, MEM_DEVICE = 0, 1
CHUNK_SCANNED, CHUNK_ALL_MATCH, CHUNK_NONE_MATCH = 0, 1, 2
INVALID_VALUE_ID = 0xFFFFFFFF
INVALID_CHUNK_ID = 0xFFFFFFFF
FOR_BLOCK_SIZE = 2048
SCAN_MATERIALIZE_ALL_MATCH = 1
SCAN_CHUNK_REGIONS = 2
POSLIST_DENSE, POSLIST_CHUNK_REGIONS = 0, 1
POOL_KEEP_MEDIAN = 1
CHUNK_DEFAULT_SIZE = 65535  # Chunk::DEFAULT_SIZE, storage/chunk.hpp:52


class RowID(C.Structure):
    _fields_ = [("chunk_id", C.c_uint32), ("chunk_offset", C.c_uint32)]


class Segment(C.Structure):
    _fields_ = [("encoding", C.c_uint32), ("data_type", C.c_uint32), ("size", C.c_uint32), ("width", C.c_uint32),
                ("data", C.c_void_p), ("aux", C.c_void_p), ("aux_size", C.c_uint32), ("ref_chunk_id", C.c_uint32),
                ("nulls", C.c_void_p), ("ref", C.c_void_p), ("sorted_by", C.c_uint32), ("bits", C.c_uint32)]


class Lz4Blocks(C.Structure):
    """hy_lz4_blocks: what an LZ4Segment<T> holds -- its blocks, compressed one by one against the dictionary."""
    _fields_ = [("blocks", C.POINTER(C.c_void_p)), ("block_bytes", C.POINTER(C.c_uint32)), ("block_count", C.c_uint32), ("block_size", C.c_uint32),
                ("last_block_size", C.c_uint32), ("dictionary_bytes", C.c_uint32), ("dictionary", C.c_void_p)]


class Value(C.Union):
    _fields_ = [("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float), ("f64", C.c_double),
                ("value_id", C.c_uint32)]


class Predicate(C.Structure):
    _fields_ = [("condition", C.c_uint32), ("value_type", C.c_uint32), ("value", Value), ("value2", Value),
                ("per_chunk_lower", C.c_void_p), ("per_chunk_upper", C.c_void_p), ("per_chunk_found", C.c_void_p),
                ("column_is_nullable", C.c_uint32), ("reserved", C.c_uint32), ("match_words", C.c_void_p), ("match_word_offsets", C.c_void_p)]


class ScanResult(C.Structure):
    _fields_ = [("mem", C.c_uint32), ("flags", C.c_uint32), ("matches", C.c_void_p), ("capacity", C.c_uint64),
                ("offsets", C.c_void_p), ("counts", C.c_void_p), ("chunk_state", C.c_void_p),
                ("total_matches", C.c_uint64)]


class JoinResult(C.Structure):
    _fields_ = [("mem", C.c_uint32), ("radix_bits", C.c_uint32), ("left_pos", C.c_void_p), ("right_pos", C.c_void_p),
                ("capacity", C.c_uint64), ("slice_offsets", C.c_void_p), ("slice_capacity", C.c_uint32),
                ("n_slices", C.c_uint32), ("n_pairs", C.c_uint64), ("left_is_build", C.c_uint32),
                ("flags", C.c_uint32), ("status", C.c_void_p)]


JOIN_ASYNC = 1


class JoinStatus(C.Structure):
    """hy_join_status: what a HY_JOIN_ASYNC join leaves in device memory."""
    _fields_ = [("n_pairs", C.c_uint64), ("n_slices", C.c_uint32), ("fits", C.c_uint32), ("build_confirmed", C.c_uint32), ("error", C.c_uint32),
                ("reserved", C.c_uint64)]


class JoinPredicate(C.Structure):
    """hy_join_predicate: left_column <condition> right_column, evaluated on the pairs the primary equality finds."""
    _fields_ = [("left_column", C.c_void_p), ("right_column", C.c_void_p), ("condition", C.c_uint32), ("reserved", C.c_uint32)]


MAX_SECONDARY_PREDICATES = 4
# hy_set_option (include/hyrise_amd.h HY_OPT_*): equivalent paths / launch shapes; every setting gives the same results
(OPT_ALLOW_ANY_ARCH, OPT_JOIN_RANK_TABLE, OPT_JOIN_HINT, OPT_JOIN_BREAK_HINT, OPT_JOIN_PKFK, OPT_JOIN_LDS_BUILD, OPT_JOIN_LDS_BUILD_TILES, OPT_JOIN_FILL_WGS_PER_CU,
 OPT_JOIN_HAND_OVER_RANKS, OPT_AGG_PARTITION_BITS, OPT_AGG_SPILL_SHIFT, OPT_AGG_SMALL_DOMAIN, OPT_FUSED_SMALL_DOMAIN, OPT_SCAN_TWO_COLUMNS, OPT_STAR_FUSED_PROBE, OPT_STAR_FUSED_FINISH) = range(16)
KERNEL_OTHER, KERNEL_SCAN, KERNEL_JOIN_PROBE, KERNEL_JOIN_COUNT, KERNEL_JOIN_BUILD, KERNEL_AGGREGATE, KERNEL_PROJECTION = range(7)   # hy_profile_read_kernel
ARITH_ADD, ARITH_SUB, ARITH_MUL, ARITH_DIV, ARITH_MOD = range(5)


class Operand(C.Structure):
    _fields_ = [("column", C.c_void_p), ("literal_type", C.c_uint32), ("literal", Value)]


class AggregateSpec(C.Structure):
    _fields_ = [("function", C.c_uint32), ("column", C.c_void_p)]


EXPR_COLUMN, EXPR_LITERAL, EXPR_ARITHMETIC = range(3)
MAX_EXPRESSION_NODES, MAX_FILTERS = 12, 4
FILTER_VALIDATE = 0x100   # hy_filter.predicate.condition of a Validate filter (hy_scan_project_aggregate)


class ExpressionNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("op", C.c_uint32), ("column", C.c_void_p), ("literal_type", C.c_uint32), ("reserved", C.c_uint32),
                ("literal", Value)]


class Expression(C.Structure):
    """hy_expression: postfix, at most three stack slots."""
    _fields_ = [("n_nodes", C.c_uint32), ("reserved", C.c_uint32), ("nodes", ExpressionNode * MAX_EXPRESSION_NODES)]


class Filter(C.Structure):
    _fields_ = [("column", C.c_void_p), ("predicate", Predicate)]


class FusedAggregate(C.Structure):
    _fields_ = [("function", C.c_uint32), ("reserved", C.c_uint32), ("input", C.POINTER(Expression))]


class AggregateColumn(C.Structure):
    _fields_ = [("data_type", C.c_uint32), ("reserved", C.c_uint32), ("values", C.c_void_p), ("is_null", C.c_void_p)]


class AggregateResult(C.Structure):
    _fields_ = [("mem", C.c_uint32), ("group_capacity", C.c_uint32), ("n_groups", C.c_uint32),
                ("reserved", C.c_uint32), ("group_row_ids", C.c_void_p), ("columns", C.POINTER(AggregateColumn))]


STAR_NO_OP = 0xFFFFFFFF
MAX_STAR_DIMENSIONS, MAX_STAR_AGGREGATES = 8, 8


class StarDimension(C.Structure):
    _fields_ = [("key", C.c_void_p), ("filter_column", C.c_void_p), ("predicate", Predicate), ("fact_key", C.c_void_p)]


class StarColumn(C.Structure):
    _fields_ = [("table", C.c_uint32), ("reserved", C.c_uint32), ("column", C.c_void_p)]


class StarAggregate(C.Structure):
    _fields_ = [("function", C.c_uint32), ("op", C.c_uint32), ("left", StarColumn), ("right", StarColumn)]


# every symbol include/hyrise_amd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("hy_abi_version", C.c_int32, []),
    ("hy_init", C.c_int32, [C.c_int32]),
    ("hy_set_option", C.c_int32, [C.c_uint32, C.c_int64]),
    ("hy_get_option", C.c_int32, [C.c_uint32, C.POINTER(C.c_int64)]),
    ("hy_shutdown", C.c_int32, []),
    ("hy_last_error", C.c_char_p, []),
    ("hy_set_stream", C.c_int32, [C.c_void_p]),
    ("hy_synchronize", C.c_int32, []),
    ("hy_device_malloc", C.c_int32, [C.POINTER(C.c_void_p), C.c_size_t]),
    ("hy_device_free", C.c_int32, [C.c_void_p]),
    ("hy_memcpy_h2d", C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("hy_memcpy_d2h", C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("hy_device_count", C.c_int32, [C.POINTER(C.c_int32)]),
    ("hy_set_profiling", C.c_int32, [C.c_int32]),
    ("hy_profile_read", C.c_int32, [C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    ("hy_profile_read_kernel", C.c_int32, [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    ("hy_profile_event_overhead", C.c_int32, [C.POINTER(C.c_float)]),
    ("hy_bind_device", C.c_int32, [C.c_int32]),
    ("hy_comm_unique_id", C.c_int32, [C.c_void_p]),
    ("hy_comm_init_rank", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("hy_comm_init_all", C.c_int32, [C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("hy_comm_destroy", C.c_int32, [C.c_void_p]),
    ("hy_comm_rank", C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("hy_comm_group_begin", C.c_int32, []),
    ("hy_comm_group_end", C.c_int32, []),
    ("hy_comm_all_reduce", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
    ("hy_comm_all_gather", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("hy_comm_all_to_all_v", C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64)]),
    ("hy_column_create", C.c_int32, [C.POINTER(Segment), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("hy_column_destroy", C.c_int32, [C.c_void_p]),
    ("hy_column_row_count", C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("hy_column_chunk_count", C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("hy_table_scan", C.c_int32, [C.c_void_p, C.POINTER(Predicate), C.c_void_p, C.c_uint32, C.POINTER(ScanResult)]),
    ("hy_table_scan_columns", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ScanResult)]),
    ("hy_projection_arithmetic", C.c_int32, [C.c_uint32, C.POINTER(Operand), C.POINTER(Operand), C.POINTER(C.c_void_p)]),
    ("hy_column_read_chunk", C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("hy_column_data_type", C.c_uint32, [C.c_void_p]),
    ("hy_column_chunk_rows", C.c_uint32, [C.c_void_p, C.c_uint32]),
    ("hy_validate", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ScanResult)]),
    ("hy_predicate_cast", C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Predicate)]),
    ("hy_join_output_chunks", C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("hy_poslist_translate", C.c_int32, [C.c_void_p, C.POINTER(ScanResult), C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("hy_join_hash", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(JoinResult)]),
    ("hy_join_hash_finish", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(JoinResult)]),
    ("hy_join_hash_predicates", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(JoinPredicate), C.c_uint32, C.POINTER(JoinResult)]),
    ("hy_join_hash_radix_bits", C.c_int32, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]),
    ("hy_join_hash_count", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("hy_aggregate_hash", C.c_int32, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(AggregateSpec), C.c_uint32,
                                      C.POINTER(AggregateResult)]),
    ("hy_star_join_aggregate", C.c_int32, [C.POINTER(StarDimension), C.c_uint32, C.POINTER(StarColumn), C.c_uint32, C.POINTER(StarAggregate), C.c_uint32,
                                           C.POINTER(AggregateResult), C.POINTER(C.c_uint64)]),
    ("hy_scan_project_aggregate", C.c_int32, [C.POINTER(Filter), C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(FusedAggregate), C.c_uint32,
                                              C.POINTER(AggregateResult)]),
    ("hy_column_export", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("hy_repartition_count", C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("hy_repartition_pack", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("hy_gather_row_ids", C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("hy_poslist_gather", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("hy_result_pool_acquire", C.c_int32, [C.c_uint64, C.POINTER(C.c_void_p)]),
    ("hy_result_pool_acquire_pair", C.c_int32, [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("hy_result_pool_release", C.c_int32, [C.c_void_p]),
    ("hy_result_pool_calibrate", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    ("hy_result_pool_trim", C.c_int32, []),
    ("hy_result_pool_stats", C.c_int32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
]


class HyriseAmdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"hyrise_amd status {status}: {message}")
        self.status = status


_lib = None


def load_library():
    """dlopen libhyrise_amd.so (built by __graft_entry__.build()); raises if it is missing -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HY_LIBRARY", LIB_PATH)   # (another build of the same library: A/B timing)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). hyrise_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status != OK:
        raise HyriseAmdError(status, load_library().hy_last_error().decode("utf-8", "replace"))


class option:
    """`with abi.option(abi.OPT_JOIN_PKFK, 0): ...` -- one of the library's options set for the block (tests force paths, tools A/B them)."""

    def __init__(self, option_id, value):
        self.option_id, self.value, self.before = option_id, value, C.c_int64(0)

    def __enter__(self):
        lib = load_library()
        check(lib.hy_get_option(self.option_id, C.byref(self.before)))
        check(lib.hy_set_option(self.option_id, self.value))
        return self

    def __exit__(self, *exc):
        check(load_library().hy_set_option(self.option_id, self.before.value))
        return False


# The switches the tools/ scripts name -> (option, value when the switch is "1" / its integer value otherwise).
_SWITCHES = {
    "HY_SCAN_NO_TWO_COLUMNS": (OPT_SCAN_TWO_COLUMNS, 0), "HY_STAR_NO_FUSED_PROBE": (OPT_STAR_FUSED_PROBE, 0), "HY_STAR_NO_FUSED_FINISH": (OPT_STAR_FUSED_FINISH, 0),
    "HY_JOIN_NO_RANK_TABLE": (OPT_JOIN_RANK_TABLE, 0), "HY_JOIN_NO_HINT": (OPT_JOIN_HINT, 0), "HY_JOIN_BREAK_HINT": (OPT_JOIN_BREAK_HINT, None),
    "HY_JOIN_NO_PKFK": (OPT_JOIN_PKFK, 0), "HY_JOIN_NO_LDS_BUILD": (OPT_JOIN_LDS_BUILD, 0), "HY_JOIN_LDS_BUILD_TILES": (OPT_JOIN_LDS_BUILD_TILES, None),
    "HY_JOIN_FILL_WGS_PER_CU": (OPT_JOIN_FILL_WGS_PER_CU, None), "HY_JOIN_HAND_OVER_RANKS": (OPT_JOIN_HAND_OVER_RANKS, None),
    "HY_AGG_PARTITION_BITS": (OPT_AGG_PARTITION_BITS, None), "HY_AGG_SPILL_SHIFT": (OPT_AGG_SPILL_SHIFT, None), "HY_AGG_NO_SMALL_DOMAIN": (OPT_AGG_SMALL_DOMAIN, 0),
    "HY_FUSED_NO_SMALL_DOMAIN": (OPT_FUSED_SMALL_DOMAIN, 0)
}


class switches:
    """`with abi.switches({"HY_JOIN_NO_PKFK": "1"}): ...` -- the tools' named switches as options for the block.  Names that are not options
    (HY_*_TRACE, HY_*_TIMING, HY_*_DEBUG) are debug aids of a -DHY_DEBUG_SWITCHES build: they go into the environment."""

    def __init__(self, named):
        self.named, self.stack, self.env = dict(named), [], []

    def __enter__(self):
        for name, value in self.named.items():
            if name in _SWITCHES:
                option_id, fixed = _SWITCHES[name]
                o = option(option_id, fixed if fixed is not None else int(value))
                o.__enter__()
                self.stack.append(o)
            else:
                self.env.append((name, os.environ.get(name)))
                os.environ[name] = str(value)
        return self

    def __exit__(self, *exc):
        for o in reversed(self.stack):
            o.__exit__(*exc)
        for name, before in self.env:
            if before is None:
                del os.environ[name]
            else:
                os.environ[name] = before
        return False
