"""The two host-side rules of the boundary (hyrise_amd/csrc/boundary.hip, plain arithmetic behind the C ABI) against their CPU
restatements in oracle/ and against the reference's own known answers:
  hy_predicate_cast      lossless_predicate_cast_test.cpp (NextFloatTowards, NonFloatTypes, FloatTypeWith*) + TableScan's use of it
  hy_join_output_chunks  write_output_chunks' MIN_SIZE / MAX_SIZE merge (join_output_writing.cpp:245-296)
No GPU is needed: neither entry point touches the device."""
import ctypes as C
import struct

import numpy as np
import pytest

from hyrise_amd import abi
from hyrise_amd.operators import _literal, join_output_chunks, predicate_for_column
from support import oracle

INT, LONG, FLOAT, DOUBLE = abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE
EQ, NE, LT, LE, GT, GE = (abi.PRED_EQUALS, abi.PRED_NOT_EQUALS, abi.PRED_LESS_THAN, abi.PRED_LESS_THAN_EQUALS, abi.PRED_GREATER_THAN,
                          abi.PRED_GREATER_THAN_EQUALS)
BETWEENS = (abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_BETWEEN_LOWER_EXCLUSIVE, abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.PRED_BETWEEN_EXCLUSIVE)
FIELD = {INT: "i32", LONG: "i64", FLOAT: "f32", DOUBLE: "f64"}


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def oracle_predicate(condition, column_type, literal_type, literal, literal2_type=None, literal2=None):
    lib = oracle()
    lib.hyo_predicate_for_column.restype = C.c_int
    lib.hyo_predicate_for_column.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(abi.Predicate)]
    first = _literal(literal_type, literal)
    second = _literal(literal2_type, literal2) if literal2 is not None else None
    out = abi.Predicate()
    ok = lib.hyo_predicate_for_column(condition, column_type, literal_type, C.addressof(first), literal2_type or abi.TYPE_NULL,
                                      C.addressof(second) if second is not None else None, C.byref(out))
    return out if ok else None


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    raw = lambda p: (p.condition, p.value_type, bytes(p.value), bytes(p.value2))
    return raw(a) == raw(b)


def test_next_float_towards_known_answers():   # lossless_predicate_cast_test.cpp:13-46
    lib = oracle()
    lib.hyo_next_float_towards.restype = C.c_int
    lib.hyo_next_float_towards.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_float)]
    big, bigger = 340282346638528859811704183484516925440.0, 340282346638528897590636046441678635008.0

    def towards(value, to):
        out = C.c_float(0)
        return out.value if lib.hyo_next_float_towards(value, to, C.byref(out)) else None

    assert towards(3, 2) == f32(2.9999997615814208984375) and towards(3, 4) == f32(3.0000002384185791015625)
    assert towards(3.1, 3) == f32(3.099999904632568359375) and towards(3.1, 4) == f32(3.1000001430511474609375)
    assert towards(3.1, 3.1) is None
    assert towards(big, 0) == f32(340282326356119256160033759537265639424.0) and towards(big, big * 10) is None
    assert towards(bigger, 0) is None and towards(bigger, bigger * 10) is None
    assert towards(-big, -10) == f32(-340282326356119256160033759537265639424.0) and towards(-big, -big * 10) is None
    assert towards(-bigger, 10) is None and towards(-bigger, -bigger * 10) is None


def test_predicate_cast_known_answers():   # lossless_predicate_cast_test.cpp:48-98
    def got(condition, column_type, literal_type, literal):
        p = predicate_for_column(condition, column_type, literal_type, literal)
        assert same(p, oracle_predicate(condition, column_type, literal_type, literal))
        return None if p is None else (p.condition, getattr(p.value, FIELD[column_type]))

    assert got(GT, LONG, LONG, 10) == (GT, 10) and got(EQ, LONG, LONG, 10) == (EQ, 10)          # input type == output type
    assert got(GT, INT, LONG, 10) == (GT, 10) and got(GT, INT, LONG, 100_000_000_000) is None      # downcast
    assert got(GT, LONG, INT, 10) == (GT, 10)                                                        # upcast
    assert got(GT, FLOAT, DOUBLE, 3.0) == (GT, 3.0)                                                  # lossless
    assert got(LT, FLOAT, DOUBLE, 3.1) == (LE, f32(3.099999904632568359375)) and got(LE, FLOAT, DOUBLE, 3.1) == (LE, f32(3.099999904632568359375))
    assert got(EQ, FLOAT, DOUBLE, 3.1) is None and got(NE, FLOAT, DOUBLE, 3.1) is None
    assert got(GT, FLOAT, DOUBLE, 3.1) == (GE, f32(3.1000001430511474609375)) and got(GE, FLOAT, DOUBLE, 3.1) == (GE, f32(3.1000001430511474609375))
    # more of lossless_cast.hpp: fractions, the integral range bounds, integers a float cannot hold
    assert got(LT, INT, DOUBLE, 3.5) is None and got(LT, INT, DOUBLE, 3.0) == (LT, 3) and got(LT, INT, FLOAT, -7.0) == (LT, -7)
    assert got(EQ, INT, DOUBLE, 2147483648.0) is None and got(EQ, INT, DOUBLE, -2147483648.0) == (EQ, -2147483648)
    assert got(EQ, LONG, DOUBLE, 9223372036854775808.0) is None and got(EQ, LONG, DOUBLE, -9223372036854775808.0) == (EQ, -9223372036854775808)
    assert got(EQ, FLOAT, INT, 16777217) is None and got(EQ, FLOAT, INT, 16777216) == (EQ, 16777216.0)
    assert got(EQ, DOUBLE, LONG, 2**53 + 1) is None and got(EQ, DOUBLE, LONG, 2**53) == (EQ, float(2**53))
    assert got(GE, DOUBLE, FLOAT, f32(0.1)) == (GE, f32(0.1))


def test_between_bounds_are_cast_one_by_one():   # table_scan.cpp:406-448
    # TPC-H Q6: l_discount BETWEEN 0.06 - 0.01 AND 0.06 + 0.01001 on a float column (tpch_queries.cpp:206-210)
    p = predicate_for_column(abi.PRED_BETWEEN_INCLUSIVE, FLOAT, DOUBLE, 0.06 - 0.01, DOUBLE, 0.06 + 0.01001)
    assert p.condition == abi.PRED_BETWEEN_INCLUSIVE and p.value_type == FLOAT
    # 0.06 - 0.01 is just below 0.05f and 0.06 + 0.01001 no float at all: the bounds move to the nearest floats INSIDE the range
    assert 0.06 - 0.01 < float(np.float32(0.05)) and p.value.f32 == f32(0.05)
    assert float(p.value2.f32) <= 0.06 + 0.01001 < float(np.nextafter(np.float32(p.value2.f32), np.float32(1)))
    # a strict bound that has no float twin becomes inclusive; one that has stays strict
    p = predicate_for_column(abi.PRED_BETWEEN_EXCLUSIVE, FLOAT, DOUBLE, 3.1, DOUBLE, 4.0)
    assert (p.condition, p.value.f32, p.value2.f32) == (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, f32(3.1000001430511474609375), 4.0)
    p = predicate_for_column(abi.PRED_BETWEEN_EXCLUSIVE, FLOAT, DOUBLE, 3.0, DOUBLE, 4.1)
    assert (p.condition, p.value.f32, p.value2.f32) == (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, 3.0, f32(4.099999904632568359375))
    assert predicate_for_column(abi.PRED_BETWEEN_INCLUSIVE, INT, LONG, 5, DOUBLE, 7.5) is None       # 7.5 is no int: ExpressionEvaluator scan
    p = predicate_for_column(abi.PRED_BETWEEN_LOWER_EXCLUSIVE, INT, LONG, 5, DOUBLE, 7.0)
    assert (p.condition, p.value.i32, p.value2.i32) == (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, 5, 7)


def test_predicate_cast_matches_the_oracle_everywhere():
    rng = np.random.default_rng(31)
    doubles = np.concatenate([rng.normal(0, 1, 200), rng.normal(0, 1e6, 200), rng.integers(-2**40, 2**40, 200).astype(np.float64), rng.normal(0, 1, 100).astype(np.float32),
                              [0.0, -0.0, 1e39, -1e39, 3.4028234663852886e38, 3.4028235e38, 2.0**31, -2.0**31, 2.0**63, -2.0**63, 16777217.0, 1e-46, 5e-324, np.inf, -np.inf]])
    longs = np.concatenate([rng.integers(-2**62, 2**62, 200), rng.integers(-2**31 - 5, 2**31 + 5, 200), [2**24, 2**24 + 1, 2**53, 2**53 + 1, 2**63 - 1, -2**63, 2**31 - 1, 2**31, -2**31, -2**31 - 1]])
    literals = [(DOUBLE, float(x)) for x in doubles] + [(FLOAT, float(np.float32(x))) for x in doubles[np.abs(doubles) < 3e38][:300]] + \
               [(LONG, int(x)) for x in longs] + [(INT, int(x)) for x in longs if -2**31 <= x < 2**31]
    checked = 0
    for literal_type, literal in literals:
        for column_type in (INT, LONG, FLOAT, DOUBLE):
            for condition in (EQ, NE, LT, LE, GT, GE):
                assert same(predicate_for_column(condition, column_type, literal_type, literal), oracle_predicate(condition, column_type, literal_type, literal)), \
                    (condition, column_type, literal_type, literal)
                checked += 1
    for _ in range(3000):
        (t1, v1), (t2, v2) = literals[rng.integers(len(literals))], literals[rng.integers(len(literals))]
        column_type, condition = (INT, LONG, FLOAT, DOUBLE)[rng.integers(4)], BETWEENS[rng.integers(4)]
        assert same(predicate_for_column(condition, column_type, t1, v1, t2, v2), oracle_predicate(condition, column_type, t1, v1, t2, v2)), (condition, column_type, t1, v1, t2, v2)
    assert checked > 20_000


def test_null_literal_and_bad_conditions():
    lib = abi.load_library()
    value, out = _literal(INT, 1), abi.Predicate()
    assert lib.hy_predicate_cast(EQ, INT, abi.TYPE_NULL, C.addressof(value), abi.TYPE_NULL, None, C.byref(out)) == abi.ERR_UNSUPPORTED
    assert lib.hy_predicate_cast(abi.PRED_IS_NULL, INT, INT, C.addressof(value), abi.TYPE_NULL, None, C.byref(out)) == abi.ERR_INVALID
    assert lib.hy_predicate_cast(abi.PRED_BETWEEN_INCLUSIVE, INT, INT, C.addressof(value), abi.TYPE_NULL, None, C.byref(out)) == abi.ERR_INVALID
    assert oracle_predicate(EQ, INT, abi.TYPE_NULL, 0) is None


# ---- write_output_chunks ----------------------------------------------------------------------------------------------------
def model_chunks(sizes):
    """join_output_writing.cpp:245-296 on plain Python lists (independent of both C versions)."""
    lists = [list(range(n)) for n in sizes]
    chunks, p = [], 0
    while p < len(lists):
        current = lists[p]
        if not current:
            p += 1
            continue
        while p + 1 < len(lists) and len(current) < 1000 and len(current) + len(lists[p + 1]) < 4000:
            current = current + lists[p + 1]
            p += 1
        chunks.append(len(current))
        p += 1
    return chunks


def oracle_chunks(offsets, merge=1):
    lib = oracle()
    lib.hyo_write_output_chunks.restype = C.c_uint32
    lib.hyo_write_output_chunks.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    out = np.zeros(len(offsets), dtype=np.uint64)
    n = lib.hyo_write_output_chunks(offsets.ctypes.data, len(offsets) - 1, merge, out.ctypes.data)
    return out[:n + 1]


@pytest.mark.parametrize("sizes", [[], [0], [0, 0, 0], [5], [999, 1], [999, 3000, 1], [999, 3001], [1000, 5], [5, 1000], [400, 400, 400, 400, 400, 400, 400, 400, 400, 400],
                                   [0, 10, 0, 0, 20, 5000, 0, 3, 0], [3999], [500, 3499, 1], [500, 3500], [131070, 1, 131070, 2, 2]])
def test_output_chunk_merge_cases(sizes):
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    want = np.concatenate([[0], np.cumsum(model_chunks(sizes))]).astype(np.uint64)
    np.testing.assert_array_equal(oracle_chunks(offsets), want)
    np.testing.assert_array_equal(join_output_chunks(offsets, len(sizes)), want)
    unmerged = np.concatenate([[0], np.cumsum([s for s in sizes if s])]).astype(np.uint64)
    np.testing.assert_array_equal(oracle_chunks(offsets, merge=0), unmerged)


def test_output_chunk_merge_random():
    rng = np.random.default_rng(2)
    for _ in range(300):
        n = int(rng.integers(1, 200))
        sizes = np.where(rng.random(n) < 0.3, 0, rng.integers(1, int(rng.choice([50, 1200, 5000, 140_000])), n)).tolist()
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        want = np.concatenate([[0], np.cumsum(model_chunks(sizes))]).astype(np.uint64)
        np.testing.assert_array_equal(oracle_chunks(offsets), want)
        np.testing.assert_array_equal(join_output_chunks(offsets, n), want)
