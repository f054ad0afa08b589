"""Pins the CPU oracle's JoinHash: the reference's step-level known answers (join_hash_steps_test.cpp,
join_hash_test.cpp), differential testing against a nested-loop JoinVerification on the reference's own
join_test_runner inputs (join_test_runner.cpp:656-791), and the ordering contract of probe()."""
import ctypes as C

import numpy as np
import pytest

from hyrise_amd import abi, storage
from support import (OracleCol, _bind_join, build_column, column_values, join_result_multiset, load_tbl, oracle,
                     oracle_join, verification_join)

MODES = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE,
         abi.JOIN_ANTI_NULL_AS_FALSE]
MODE_IDS = ["Inner", "Left", "Right", "Semi", "AntiNullAsTrue", "AntiNullAsFalse"]


def test_radix_bit_calculation():
    """join_hash_test.cpp:122-129 + the SF10 value derived in SURVEY.md 8(a) A10."""
    lib = oracle()
    _bind_join(lib)
    assert lib.hyo_calculate_radix_bits(1, 0) == 0
    assert lib.hyo_calculate_radix_bits(0, 1) == 0
    assert lib.hyo_calculate_radix_bits(0, 0) == 0
    assert lib.hyo_calculate_radix_bits(1, 1) == 0
    assert lib.hyo_calculate_radix_bits(2**64 - 1, 2**64 - 1) > 0
    assert lib.hyo_calculate_radix_bits(15_000_000, 59_986_052) == 7


def materialize(column, keep_nulls, radix_bits, bloom_in=None):
    lib = oracle()
    _bind_join(lib)
    col = OracleCol(column)
    n = column.rows
    rows = np.zeros((max(1, n), 2), dtype=np.uint32)
    values = np.zeros(max(1, n), dtype=np.int64)
    nulls = np.zeros(max(1, n), dtype=np.uint8)
    counts = np.zeros(max(1, column.n_chunks), dtype=np.uint64)
    hist = np.zeros(max(1, column.n_chunks) << radix_bits, dtype=np.uint64)
    bloom_out = np.zeros(16384, dtype=np.uint64)
    total = lib.hyo_join_materialize(C.byref(col.c), int(keep_nulls), radix_bits,
                                     bloom_in.ctypes.data if bloom_in is not None else None, bloom_out.ctypes.data,
                                     rows.ctypes.data, values.ctypes.data, nulls.ctypes.data, counts.ctypes.data,
                                     hist.ctypes.data)
    return rows[:total], values[:total], nulls[:total], counts, hist.reshape(max(1, column.n_chunks), -1), bloom_out


def bloom_bits(words):
    return set(np.nonzero(np.unpackbits(words.view(np.uint8), bitorder="little"))[0].tolist())


def test_hashed_type_of_the_reference():
    """join_hash_traits_test.cpp (IntegerTraits, FloatingTraits, MixedNumberTraits): the type both sides are cast to before hashing."""
    import ctypes as C
    I, L, F, D = abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE
    hashed = oracle().hyo_join_hashed_type
    hashed.restype, hashed.argtypes = C.c_uint32, [C.c_uint32, C.c_uint32]
    table = {(I, I): I, (L, L): L, (I, L): L, (L, I): L, (F, F): F, (D, D): D, (F, D): D, (D, F): D, (I, F): F, (F, I): F, (I, D): D, (D, I): D,
             (L, F): F, (F, L): F, (L, D): D, (D, L): D}
    for (left, right), expected in table.items():
        assert hashed(left, right) == expected, (left, right)


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64], ids=lambda t: t.__name__)
def test_hash_map_finds_every_row_of_the_reference_inputs(np_type):
    """join_hash_types_test.cpp:57-76 (BuildSingleValueLargePosList: 500 x the value 17; BuildSingleRowIds: i^3 for i < 500) builds the
    hash map and looks every row up by its key.  As a join of the column with itself: every row pairs with exactly the rows of its
    key -- 500 x 500 pairs, and 500 pairs (i, i)."""
    same = build_column(np.full(500, 17).astype(np_type), None, 500, abi.ENC_UNENCODED)
    result = oracle_join(same, same, abi.JOIN_INNER, capacity=500 * 500 + 16)
    assert result.n_pairs == 500 * 500
    pairs = set(zip(result.left[:result.n_pairs, 1].tolist(), result.right[:result.n_pairs, 1].tolist()))
    assert len(pairs) == 500 * 500
    cubes = build_column((np.arange(500, dtype=np.float64) ** 3).astype(np_type), None, 500, abi.ENC_UNENCODED)
    result = oracle_join(cubes, cubes, abi.JOIN_INNER)
    n = result.n_pairs
    assert n == 500 and sorted(result.left[:n, 1].tolist()) == list(range(500))
    assert (result.left[:n, 1] == result.right[:n, 1]).all()


def test_materialize_bloom_filters():
    """join_hash_steps_test.cpp:169-220 on int_int4_with_null.tbl (chunk size 10)."""
    values = np.array([18, 7, 7, 9, 6, 0, 13, 0, 9, 7, 0], dtype=np.int32)      # column a, NULLs at rows 5 and 7
    nulls = np.zeros(11, dtype=bool)
    nulls[[5, 7]] = True
    column = build_column(values, nulls, 10, abi.ENC_UNENCODED)
    _, _, _, _, _, bloom = materialize(column, False, 1)
    assert bloom_bits(bloom) == {0, 6, 7, 9, 13, 18}                             # :169-188
    bloom_in = np.zeros(16384, dtype=np.uint64)
    for v in (6, 7, 9):
        bloom_in[v // 64] |= np.uint64(1) << np.uint64(v % 64)
    rows, vals, _, _, _, _ = materialize(column, False, 1, bloom_in)
    assert vals.tolist() == [7, 7, 9, 6, 9, 7]                                   # :190-220
    assert rows[:, 1].tolist() == [1, 2, 3, 4, 8, 9]


def test_materialize_histograms():
    """join_hash_steps_test.cpp:222-263: 1000 rows of i % 2 in chunks of 10."""
    column = build_column((np.arange(1000) % 2).astype(np.int32), None, 10, abi.ENC_UNENCODED)
    _, _, _, _, hist, _ = materialize(column, False, 1)
    assert hist.shape == (100, 2) and np.all(hist == 5)
    _, _, _, _, hist2, _ = materialize(column, False, 2)
    assert np.all((hist2 == 5) | (hist2 == 0)) and int((hist2 == 0).sum()) == 200


def runner_tables(size, chunk):
    left = load_tbl(f"join_test_runner/input_table_left_{size}.tbl")
    right = load_tbl(f"join_test_runner/input_table_right_{size}.tbl")
    return left, right


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_join_against_verification_on_runner_tables(mode):
    """join_test_runner.cpp: every join mode x table sizes {0,10,15} x chunk sizes {10,3,1} x nullable int columns,
    several radix settings; results compared as multisets like the reference does (:786-791)."""
    for lsize in (0, 10, 15):
        for rsize in (0, 10, 15):
            lt, rt = runner_tables(lsize, 0)[0], runner_tables(rsize, 0)[1]
            for column_name in ("int", "int_null", "long", "long_null"):
                lvals, lnull = lt.column("l_" + column_name)
                rvals, rnull = rt.column("r_" + column_name)
                for chunk in (10, 3, 1):
                    for encoding in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
                        left = build_column(lvals, lnull, chunk, encoding)
                        right = build_column(rvals, rnull, chunk, encoding)
                        for radix_bits in (None, 0, 1, 2, 5):
                            got = oracle_join(left, right, mode, radix_bits)
                            assert join_result_multiset(got, mode) == verification_join(left, right, mode), \
                                f"{column_name} sizes {lsize},{rsize} chunk {chunk} enc {encoding} radix {radix_bits}"


def test_join_order_contract():
    """What no reference test pins but the reference code defines (join_hash_steps.hpp:541-591,655-760): pairs come
    partition by partition (low radix bits of the key), inside a partition by probe row, inside a probe row by build
    row (insertion order)."""
    rng = np.random.default_rng(5)
    build_values = rng.integers(0, 50, 400).astype(np.int32)
    probe_values = rng.integers(0, 60, 3000).astype(np.int32)
    left = build_column(build_values, None, 64, abi.ENC_UNENCODED)      # smaller -> build side
    right = build_column(probe_values, None, 500, abi.ENC_UNENCODED)
    for radix_bits in (0, 3):
        got = oracle_join(left, right, abi.JOIN_INNER, radix_bits)
        assert got.c.left_is_build == 1
        build_ids, probe_ids = got.pairs()
        keys = [(int(probe_values[c * 500 + o]) & ((1 << radix_bits) - 1), int(c), int(o), int(bc), int(bo))
                for (bc, bo), (c, o) in zip(build_ids.tolist(), probe_ids.tolist())]
        assert keys == sorted(keys)
        assert got.n_pairs == sum(int((build_values == v).sum()) for v in probe_values)


def test_probe_slices_of_131070_elements():
    """probe() cuts every partition into slices of PROBE_SIZE_PER_CHUNK = 2 * 65535 materialised probe elements
    (join_hash_steps.hpp:47,655-660)."""
    n = 300_000
    probe_values = (np.arange(n) % 1000).astype(np.int32)
    build_values = np.arange(0, 1000, 2, dtype=np.int32)
    got = oracle_join(build_column(build_values, None, 1000, abi.ENC_UNENCODED),
                      build_column(probe_values, None, 65535, abi.ENC_UNENCODED), abi.JOIN_INNER, 0)
    # radix_bits 0: partitions == probe chunks (65535 rows each => one slice per chunk)
    assert got.c.n_slices == 5
    got1 = oracle_join(build_column(build_values, None, 1000, abi.ENC_UNENCODED),
                       build_column(probe_values, None, 65535, abi.ENC_UNENCODED), abi.JOIN_INNER, 1)
    # the probe side is filtered by the build side's Bloom filter (only even values survive): all 150 000 elements fall
    # into partition 0 => 2 slices
    assert got1.c.n_slices == 2
    offsets = got1.slice_offsets[:3].tolist()
    assert offsets == [0, 131070, 150000]


SECONDARY_SETS = [   # join_test_runner.cpp:207-211: the secondary predicate compares the first columns of the two tables ...
    [("int", abi.PRED_LESS_THAN, "int")], [("int", abi.PRED_GREATER_THAN_EQUALS, "int")], [("int", abi.PRED_NOT_EQUALS, "int")],
    # ... and, beyond the reference's sets: nullable columns, mixed types (compared in the common C++ type), two predicates
    [("int_null", abi.PRED_LESS_THAN_EQUALS, "long_null")], [("float", abi.PRED_GREATER_THAN, "long")], [("double_null", abi.PRED_EQUALS, "float_null")],
    [("int", abi.PRED_NOT_EQUALS, "int"), ("double", abi.PRED_LESS_THAN, "double")],
]
SECONDARY_MODES = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_FALSE]


def secondary_columns(lt, rt, predicate_set, chunk, encoding):
    out = []
    for left_name, condition, right_name in predicate_set:
        lvals, lnull = lt.column("l_" + left_name)
        rvals, rnull = rt.column("r_" + right_name)
        out.append((build_column(lvals, lnull, chunk, encoding), condition, build_column(rvals, rnull, chunk, encoding)))
    return out


@pytest.mark.parametrize("mode", SECONDARY_MODES)
def test_join_with_secondary_predicates_against_verification(mode):
    """Multi-predicate joins (join_test_runner.cpp:464-480): the primary equality on a key column with few distinct values,
    secondary predicates on other columns of the two tables, every join mode JoinHash supports with them, against the
    nested loops of JoinVerification -- as multisets, like the reference compares (:786-791)."""
    for lsize, rsize in ((10, 15), (15, 10), (15, 15), (0, 10), (10, 0)):
        lt, rt = runner_tables(lsize, 0)[0], runner_tables(rsize, 0)[1]
        for key in ("int", "long_null"):
            lvals, lnull = lt.column("l_" + key)
            rvals, rnull = rt.column("r_" + key)
            for chunk, encoding in ((10, abi.ENC_UNENCODED), (3, abi.ENC_DICTIONARY)):
                left, right = build_column(lvals, lnull, chunk, encoding), build_column(rvals, rnull, chunk, encoding)
                for predicate_set in SECONDARY_SETS:
                    secondary = secondary_columns(lt, rt, predicate_set, chunk, encoding)
                    for radix_bits in (None, 2):
                        got = oracle_join(left, right, mode, radix_bits, secondary=secondary)
                        assert join_result_multiset(got, mode) == verification_join(left, right, mode, secondary), \
                            f"key {key} sizes {lsize},{rsize} chunk {chunk} predicates {predicate_set} radix {radix_bits}"


def test_secondary_predicates_keep_the_join_order_and_reject_anti_null_as_true():
    """With secondary predicates the surviving pairs keep the order of the equi-join (they are a filter inside probe());
    AntiNullAsTrue with secondary predicates is unsupported (join_hash.cpp:39-44)."""
    lt, rt = runner_tables(15, 0)[0], runner_tables(15, 0)[1]
    lvals, lnull = lt.column("l_int")
    rvals, rnull = rt.column("r_int")
    left, right = build_column(lvals, lnull, 3, abi.ENC_UNENCODED), build_column(rvals, rnull, 3, abi.ENC_UNENCODED)
    secondary = secondary_columns(lt, rt, [("double", abi.PRED_LESS_THAN, "double")], 3, abi.ENC_UNENCODED)
    plain = oracle_join(left, right, abi.JOIN_INNER, 1)
    filtered = oracle_join(left, right, abi.JOIN_INNER, 1, secondary=secondary)
    pairs = [(tuple(l), tuple(r)) for l, r in zip(plain.left[:plain.n_pairs].tolist(), plain.right[:plain.n_pairs].tolist())]
    kept = [(tuple(l), tuple(r)) for l, r in zip(filtered.left[:filtered.n_pairs].tolist(), filtered.right[:filtered.n_pairs].tolist())]
    assert 0 < len(kept) < len(pairs)
    iterator = iter(pairs)
    assert all(pair in iterator for pair in kept)   # a subsequence
    from support import HostJoinResult, OracleCol, _bind_join, oracle
    import ctypes as C
    lib = oracle()
    _bind_join(lib)
    result = HostJoinResult(1000, 100)
    lcol, rcol = OracleCol(left), OracleCol(right)
    l2, r2 = OracleCol(secondary[0][0]), OracleCol(secondary[0][2])
    predicates = (abi.JoinPredicate * 1)()
    predicates[0].left_column, predicates[0].right_column, predicates[0].condition = C.addressof(l2.c), C.addressof(r2.c), abi.PRED_LESS_THAN
    assert lib.hyo_join_hash_predicates(C.byref(lcol.c), C.byref(rcol.c), abi.JOIN_ANTI_NULL_AS_TRUE, predicates, 1, C.byref(result.c), 1) == abi.ERR_UNSUPPORTED


# ---- float / double keys, also mixed with integers (JoinHashTraits) ---------------------------------------------------------
NUMERIC_COLUMNS = ("int", "int_null", "long", "long_null", "float", "float_null", "double", "double_null")


def test_std_hash_restatement_against_the_installed_libstdcxx(tmp_path):
    """hyo_std_hash (oracle/join.c: libstdc++'s _Hash_bytes, 0 for +-0.0) against std::hash<float> / std::hash<double> of
    the g++ installed here -- the radix partition, and with it the pair order, of float-keyed joins hangs on it."""
    import subprocess
    source = tmp_path / "std_hash.cpp"
    source.write_text('''#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
int main() {
  unsigned type; unsigned long long bits;
  while (std::scanf("%u %llx", &type, &bits) == 2) {
    unsigned long long hash;
    if (type == 3) { float f; const uint32_t b = static_cast<uint32_t>(bits); std::memcpy(&f, &b, 4); hash = std::hash<float>{}(f); }
    else { double d; std::memcpy(&d, &bits, 8); hash = std::hash<double>{}(d); }
    std::printf("%llx\\n", hash);
  }
}
''')
    binary = tmp_path / "std_hash"
    subprocess.check_call(["g++", "-O1", "-o", str(binary), str(source)])
    rng = np.random.default_rng(3)
    doubles = np.concatenate([np.array([0.0, -0.0, 1.0, -1.0, 1338.0, 4294968633.0, 0.1, np.inf, -np.inf, 1e-310, 724.3]), rng.normal(size=200) * 1e6,
                              rng.integers(-10**6, 10**6, 100).astype(np.float64)])
    floats = doubles.astype(np.float32)
    cases = [(abi.TYPE_DOUBLE, int(b)) for b in doubles.view(np.uint64)] + [(abi.TYPE_FLOAT, int(b)) for b in floats.view(np.uint32)]
    lines = "".join(f"{t} {b:x}\n" for t, b in cases)
    want = [int(x, 16) for x in subprocess.run([str(binary)], input=lines.encode(), stdout=subprocess.PIPE, check=True).stdout.split()]
    lib = oracle()
    lib.hyo_std_hash.restype = C.c_uint64
    lib.hyo_std_hash.argtypes = [C.c_int64, C.c_uint32]
    assert len(want) == len(cases)
    for (t, b), w in zip(cases, want):
        negative_zero = b == (1 << 63) if t == abi.TYPE_DOUBLE else b == (1 << 31)
        key = 0 if negative_zero else (b - (1 << 64) if b >= (1 << 63) else b)   # the oracle's key: the bits, -0.0 as +0.0
        assert lib.hyo_std_hash(key, t) == w, (t, hex(b))
    assert lib.hyo_std_hash(-5, abi.TYPE_LONG) == (1 << 64) - 5 and lib.hyo_std_hash(7, abi.TYPE_INT) == 7


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_join_on_all_numeric_type_pairs_of_the_runner_tables(mode):
    """join_test_runner.cpp:183-193 joins every data type with every other one: both sides are cast to JoinHashTraits'
    HashedType (int x float -> float, long x double -> double ...) and compared there, like C++ compares the two types."""
    lt, rt = runner_tables(15, 0)[0], runner_tables(10, 0)[1]
    for left_name in NUMERIC_COLUMNS:
        for right_name in NUMERIC_COLUMNS:
            if "float" not in left_name + right_name and "double" not in left_name + right_name:
                continue   # (the integer pairs: test_join_against_verification_on_runner_tables)
            lvals, lnull = lt.column("l_" + left_name)
            rvals, rnull = rt.column("r_" + right_name)
            for chunk, encoding, radix_bits in ((10, abi.ENC_UNENCODED, None), (3, abi.ENC_DICTIONARY, 2), (4, abi.ENC_DICTIONARY, 0)):
                left = build_column(lvals, lnull, chunk, encoding)
                right = build_column(rvals, rnull, chunk, encoding)
                got = oracle_join(left, right, mode, radix_bits)
                assert join_result_multiset(got, mode) == verification_join(left, right, mode), f"{left_name} x {right_name} chunk {chunk} radix {radix_bits}"


FLOAT_KEY_DOMAIN = [0.0, -0.0, 1.0, -1.0, 0.5, 16777216.0, 16777217.0, 16777218.0, 1e30, -1e30, float("inf"), float("-inf"), float("nan"), 3.25, 1338.0]


def float_key_column(rng, data_type, n, chunk, null_fraction=0.15):
    """Keys that collide after the cast to the HashedType: 16777217 is not a float, +-0.0 are one key, NaN is none."""
    np_type = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}[data_type]
    if data_type in (abi.TYPE_INT, abi.TYPE_LONG):
        domain = np.array([0, 1, -1, 16777216, 16777217, 16777218, 3, 1338], dtype=np_type)
    else:
        domain = np.array(FLOAT_KEY_DOMAIN, dtype=np_type)
    values = domain[rng.integers(0, len(domain), n)]
    nulls = rng.random(n) < null_fraction
    encoding = abi.ENC_UNENCODED if (np_type in (np.float32, np.float64) and np.isnan(values).any()) or rng.random() < 0.5 else abi.ENC_DICTIONARY
    return build_column(values, nulls, chunk, encoding)   # (a dictionary cannot hold NaN: it is not ordered)


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_float_key_joins_against_verification(mode):
    rng = np.random.default_rng(17)
    types = (abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE)
    for left_type in types:
        for right_type in types:
            if left_type in (abi.TYPE_INT, abi.TYPE_LONG) and right_type in (abi.TYPE_INT, abi.TYPE_LONG):
                continue
            for n_left, n_right, chunk, radix_bits in ((40, 25, 7, 3), (12, 60, 100, None), (30, 30, 5, 0)):
                left = float_key_column(rng, left_type, n_left, chunk)
                right = float_key_column(rng, right_type, n_right, chunk)
                got = oracle_join(left, right, mode, radix_bits)
                assert join_result_multiset(got, mode) == verification_join(left, right, mode), f"{left_type} x {right_type} rows {n_left},{n_right} radix {radix_bits}"


# ---- string keys: ids made by the adapter (hyrise_amd/join_keys.py), joined as int64 -------------------------------------------
def string_key_columns(lvals, lnull, rvals, rnull, chunk):
    """-> (left, right) int64 HostColumns over the strings' join ids, and the same columns over independent codes."""
    from hyrise_amd.join_keys import StringJoinKeys
    keys = StringJoinKeys()
    codes = {}
    out = []
    for values, nulls in ((lvals, lnull), (rvals, rnull)):
        raw = [v.encode() for v in values]
        segments, dictionaries = [], []
        for begin in range(0, len(raw), chunk):
            segment, dictionary = storage.encode_string_dictionary(raw[begin:begin + chunk], None if nulls is None else nulls[begin:begin + chunk])
            segments.append(segment)
            dictionaries.append(dictionary)
        independent = np.array([codes.setdefault(v, len(codes)) for v in raw], dtype=np.int64)
        out.append((keys.column(segments, dictionaries), build_column(independent, nulls, chunk, abi.ENC_UNENCODED)))
    return out[0][0], out[1][0], out[0][1], out[1][1]


def test_std_hash_of_strings_against_the_installed_libstdcxx(tmp_path):
    import subprocess
    from hyrise_amd.join_keys import std_hash_bytes
    source = tmp_path / "string_hash.cpp"
    source.write_text('''#include <cstdio>
#include <functional>
#include <iostream>
#include <string>
int main() { std::string line; while (std::getline(std::cin, line)) std::printf("%llx\\n", static_cast<unsigned long long>(std::hash<std::string>{}(line))); }
''')
    binary = tmp_path / "string_hash"
    subprocess.check_call(["g++", "-O1", "-o", str(binary), str(source)])
    rng = np.random.default_rng(4)
    cases = [b"", b"a", b"m", b"abcdefg", b"abcdefgh", b"abcdefghi", b"Dampfschifffahrtsgesellschaft", "kapit\u00e4n".encode()]
    cases += [bytes(rng.integers(32, 127, int(n)).astype(np.uint8)) for n in rng.integers(0, 70, 200)]
    want = [int(x, 16) for x in subprocess.run([str(binary)], input=b"".join(c + b"\n" for c in cases), stdout=subprocess.PIPE, check=True).stdout.split()]
    lib = oracle()
    lib.hyo_std_hash_bytes.restype = C.c_uint64
    lib.hyo_std_hash_bytes.argtypes = [C.c_char_p, C.c_uint64]
    assert len(want) == len(cases)
    for c, w in zip(cases, want):
        assert std_hash_bytes(c) == w and lib.hyo_std_hash_bytes(c, len(c)) == w, c


@pytest.mark.parametrize("mode", MODES, ids=MODE_IDS)
def test_string_key_joins_on_runner_tables(mode):
    """The join over the adapter's string ids is the join over the strings: same pairs as the nested loop over independent
    codes, and in radix-partition order of std::hash(string) (the low bits of an id ARE the hash's)."""
    from hyrise_amd.join_keys import std_hash_bytes
    lt, rt = runner_tables(15, 0)[0], runner_tables(10, 0)[1]
    for name in ("string", "string_null"):
        lvals, lnull = lt.column("l_" + name)
        rvals, rnull = rt.column("r_" + name)
        for chunk, radix_bits in ((10, None), (3, 2), (4, 0), (1, 8)):
            left, right, left_codes, right_codes = string_key_columns(lvals, lnull, rvals, rnull, chunk)
            got = oracle_join(left, right, mode, radix_bits)
            assert join_result_multiset(got, mode) == verification_join(left_codes, right_codes, mode), f"{name} chunk {chunk} radix {radix_bits}"
            bits = got.c.radix_bits
            if bits and mode == abi.JOIN_INNER:
                side, values = (got.right, rvals) if got.c.left_is_build else (got.left, lvals)   # the probe side's strings, in output order
                partitions = [std_hash_bytes(values[int(c) * chunk + int(o)].encode()) & ((1 << bits) - 1) for c, o in side[:got.n_pairs]]
                assert partitions == sorted(partitions) and got.n_pairs > 0
