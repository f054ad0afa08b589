"""Pins the CPU restatement of Projection arithmetic (oracle/projection.c) against the reference's known answers:
ExpressionEvaluatorToValuesTest.ArithmeticsLiterals / ArithmeticsSeries (expression_evaluator_to_values_test.cpp:228-256,
table_a = resources/test_data/tbl/expression_evaluator/input_a.tbl) and expression_common_type (expression_utils.cpp:172-204)."""
import numpy as np

from hyrise_amd import abi
from support import load_tbl, oracle, oracle_arithmetic

I, L, F, D = abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE


def one(op, left, right):
    values, nulls = oracle_arithmetic(op, left, right, n=1)
    return None if nulls[0] else values[0].item()


def test_arithmetics_literals():   # :228-242 (C++ literal types: 5 -> int32, 10.0 -> double, 23.25 -> double)
    assert one(abi.ARITH_MUL, (I, 5), (I, 3)) == 15
    assert one(abi.ARITH_MUL, (I, 5), None) is None
    assert one(abi.ARITH_ADD, (I, 5), (I, 6)) == 11
    assert one(abi.ARITH_SUB, (I, 15), (I, 12)) == 3
    assert one(abi.ARITH_DIV, (D, 10.0), (D, 4.0)) == 2.5
    assert one(abi.ARITH_DIV, (D, 10.0), (I, 0)) is None
    assert one(abi.ARITH_DIV, (I, 10), (I, 0)) is None
    assert one(abi.ARITH_MOD, (I, 5), (I, 3)) == 2
    assert one(abi.ARITH_MOD, (D, 23.25), (I, 3)) == 2.25
    assert one(abi.ARITH_MOD, (D, 23.25), (I, 0)) is None
    assert one(abi.ARITH_MOD, (I, 5), (I, 0)) is None


def test_arithmetics_series():   # :244-256
    t = load_tbl("expression_evaluator/input_a.tbl")
    a, b, c = (t.column(name) for name in "abc")

    def series(op, left, right):
        values, nulls = oracle_arithmetic(op, left, right)
        return [None if n else v.item() for v, n in zip(values, nulls)]

    assert series(abi.ARITH_MUL, a, b) == [2, 6, 12, 20]
    assert series(abi.ARITH_MOD, b, a) == [0, 1, 1, 1]
    assert series(abi.ARITH_MOD, a, c) == [1, None, 3, None]
    b_plus_c = oracle_arithmetic(abi.ARITH_ADD, b, c)
    assert series(abi.ARITH_ADD, a, b_plus_c) == [36, None, 41, None]
    assert series(abi.ARITH_ADD, a, None) == [None] * 4
    assert series(abi.ARITH_ADD, (np.zeros(0, np.int32), None), (np.zeros(0, np.int32), None)) == []


def test_projection_of_int_float_add():   # projection_test.cpp:59-64 (ExecutedOnAllChunks): int_float.tbl's a + b
    t, expected = load_tbl("int_float.tbl"), load_tbl("projection/int_float_add.tbl")
    values, nulls = oracle_arithmetic(abi.ARITH_ADD, t.column("a"), t.column("b"))
    assert values.dtype == np.float32 and not nulls.any()
    want = expected.columns[0]
    assert expected.types[0] == abi.TYPE_FLOAT and len(values) == len(want)
    # (the reference compares tables cell by cell with a float tolerance and ignores the row order, check_table_equal.cpp:34,109-115)
    assert np.allclose(np.sort(values), np.sort(want), rtol=1e-6)
    assert values.tolist() == (t.columns[0].astype(np.float32) + t.columns[1]).tolist()   # int + float computes in float


def test_expression_common_type():   # expression_utils.cpp:172-204
    common = oracle().hyo_expression_common_type
    table = {(I, I): I, (I, L): L, (L, I): L, (I, F): F, (F, I): F, (L, F): D, (F, L): D, (I, D): D, (D, F): D, (F, F): F, (L, L): L, (L, D): D,
             (abi.TYPE_NULL, L): L, (F, abi.TYPE_NULL): F}
    for (lhs, rhs), expected in table.items():
        assert common(lhs, rhs) == expected, (lhs, rhs)


def test_compute_type_is_the_cxx_common_type():
    """+ - * run in std::common_type_t<A, B> and are then cast to the result type: int64 with float computes in FLOAT and
    returns a double (expression_functors.hpp:142-143 with expression_utils.cpp:189-195)."""
    big = np.array([16_777_217], dtype=np.int64)            # not representable in float
    values, _ = oracle_arithmetic(abi.ARITH_ADD, (big, None), (F, 0.0))
    assert values.dtype == np.float64 and values[0] == 16_777_216.0
    ints = np.array([2_000_000_000], dtype=np.int32)
    wrapped, _ = oracle_arithmetic(abi.ARITH_ADD, (ints, None), (ints, None))   # int32 arithmetic wraps
    assert wrapped.dtype == np.int32 and wrapped[0] == np.int32(-294967296)
    promoted, _ = oracle_arithmetic(abi.ARITH_ADD, (ints, None), (ints.astype(np.int64), None))
    assert promoted.dtype == np.int64 and promoted[0] == 4_000_000_000
    assert one(abi.ARITH_DIV, (I, 7), (I, 2)) == 3 and one(abi.ARITH_DIV, (I, -7), (I, 2)) == -3          # truncation
    assert one(abi.ARITH_DIV, (I, 7), (F, 2.0)) == 3.5
    assert one(abi.ARITH_MOD, (F, 5.5), (F, 2.0)) == 1.5 and one(abi.ARITH_MOD, (I, -7), (I, 3)) == -1
