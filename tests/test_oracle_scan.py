"""Pins the CPU oracle's TableScan against the reference's own fixtures and known-answer tests
(src/test/lib/operators/table_scan_test.cpp, table_scan_between_test.cpp) and against an independent numpy
evaluation.  CPU only."""
import numpy as np
import pytest

from golden import known_answers as KA
from hyrise_amd import abi, storage
from hyrise_amd.operators import make_predicate
from support import (brute_force_scan, build_column, decode_rows, expected_result_from_mask, load_tbl, oracle_scan,
                     oracle_scan_columns, result_rows)

ENCODINGS = [abi.ENC_UNENCODED, abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE]
ENC_IDS = ["Unencoded", "Dictionary", "FrameOfReference"]


def values_at(result, other_column):
    return decode_rows(other_column, result_rows(result))


def shuffled_tables(encoding):
    """_int_int_compressed (chunk 7, all chunks encoded) and _int_int_partly_compressed (chunk 5, chunks 0-1 encoded),
    table_scan_test.cpp:44-61."""
    out = []
    for path, chunk, enc in (("int_int_shuffled.tbl", 7, [encoding, encoding]), ("int_int_shuffled_2.tbl", 5, [encoding, encoding])):
        t = load_tbl(path)
        a = build_column(t.columns[0], None, chunk, enc)
        b = build_column(t.columns[1], None, chunk, enc)
        out.append((a, b))
    return out


@pytest.mark.parametrize("encoding", ENCODINGS, ids=ENC_IDS)
@pytest.mark.parametrize("answers,literal", [(KA.SCAN_ON_COMPRESSED_SEGMENTS, 6), (KA.SCAN_VALUE_GREATER_THAN_MAX, 30),
                                             (KA.SCAN_VALUE_LESS_THAN_MIN, -10)])
def test_scan_on_compressed_segments(encoding, answers, literal):
    for a, b in shuffled_tables(encoding):
        for condition, expected in answers.items():
            result = oracle_scan(a, make_predicate(condition, abi.TYPE_INT, literal))
            assert sorted(values_at(result, b)) == sorted(expected), f"condition {condition}"


@pytest.mark.parametrize("encoding", ENCODINGS, ids=ENC_IDS)
def test_scan_on_referenced_compressed_segments(encoding):
    """table_scan_test.cpp:433-463: scan a reference table produced by a first scan."""
    for a, b in shuffled_tables(encoding):
        first = oracle_scan(b, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 108), flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        # output of the first TableScan: one reference chunk per input chunk with matches (table_scan.cpp:199-211)
        pos_lists, single = [], []
        for c in range(b.n_chunks):
            if first.counts[c] == 0:
                continue
            pos_lists.append(first.pos_list(c).copy())
            single.append(c)
        ref_a = storage.make_reference_column(a, pos_lists, single)
        for condition, expected in KA.SCAN_ON_REFERENCED_COMPRESSED_SEGMENTS.items():
            second = oracle_scan(ref_a, make_predicate(condition, abi.TYPE_INT, 4))
            # translate positions through the input pos lists (table_scan.cpp:185-190)
            rows = [tuple(pos_lists[c][i]) for c, i in result_rows(second)]
            assert sorted(decode_rows(b, rows)) == sorted(expected), f"condition {condition}"


@pytest.mark.parametrize("encoding", ENCODINGS, ids=ENC_IDS)
def test_scan_weird_pos_list(encoding):
    """table_scan_test.cpp:126-152,465-484: one reference chunk whose pos list jumps between three chunks."""
    t = load_tbl("int_int_shuffled_2.tbl")
    a = build_column(t.columns[0], None, 5, [encoding, encoding])
    b = build_column(t.columns[1], None, 5, [encoding, encoding])
    pos = np.array(KA.WEIRD_POS_LIST, dtype=np.uint32)
    ref_a = storage.make_reference_column(a, [pos], [None])
    for condition, expected in KA.SCAN_WEIRD_POS_LIST.items():
        result = oracle_scan(ref_a, make_predicate(condition, abi.TYPE_INT, 10))
        rows = [tuple(pos[i]) for _, i in result_rows(result)]
        assert sorted(decode_rows(b, rows)) == sorted(expected), f"condition {condition}"
        # order: by referenced chunk, then original position (abstract_dereferenced_column_table_scan_impl.cpp:58-86)
        chunk_ids = [int(r[0]) for r in rows]
        assert chunk_ids == sorted(chunk_ids)


@pytest.mark.parametrize("entries,literal,expected_rows", KA.WIDE_DICTIONARY)
def test_scan_on_wide_dictionary_segment(entries, literal, expected_rows):
    values = np.arange(entries + 1, dtype=np.int32)
    column = build_column(values, None, 100_000, abi.ENC_DICTIONARY)
    assert column.segments[0].width == (2 if entries < 65536 else 4)
    result = oracle_scan(column, make_predicate(abi.PRED_GREATER_THAN, abi.TYPE_INT, literal))
    assert result.total == expected_rows


@pytest.mark.parametrize("encoding", ENCODINGS, ids=ENC_IDS)
def test_scan_for_null_values(encoding):
    t = load_tbl("int_int_w_null_8_rows.tbl")
    a = build_column(t.columns[0], t.nulls[0], 4, encoding)
    b = build_column(t.columns[1], t.nulls[1], 4, encoding)
    for condition, expected in KA.SCAN_FOR_NULL_VALUES.items():
        result = oracle_scan(b, make_predicate(condition, abi.TYPE_INT, nullable=True))
        got = values_at(result, a)
        assert sorted(got, key=lambda v: (v is None, v)) == sorted(expected, key=lambda v: (v is None, v))


@pytest.mark.parametrize("encoding", ENCODINGS, ids=ENC_IDS)
def test_single_and_double_scan_against_tbl(encoding):
    """SingleScan / DoubleScan (table_scan_test.cpp:296-302,330-342): expected tables are .tbl fixtures."""
    t = load_tbl("int_float.tbl")
    a = build_column(t.columns[0], None, 2, encoding)
    b = build_column(t.columns[1], None, 2, abi.ENC_DICTIONARY if encoding != abi.ENC_UNENCODED else encoding)
    expected = load_tbl("int_float_filtered2.tbl")
    first = oracle_scan(a, make_predicate(abi.PRED_GREATER_THAN_EQUALS, abi.TYPE_INT, 1234))
    rows = result_rows(first)
    assert sorted(decode_rows(a, rows)) == sorted(expected.columns[0].tolist())
    assert sorted(decode_rows(b, rows)) == pytest.approx(sorted(expected.columns[1].tolist()))
    expected2 = load_tbl("int_float_filtered.tbl")
    second = oracle_scan(b, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, 457.9))
    both = set(rows) & set(result_rows(second))
    assert sorted(decode_rows(a, sorted(both))) == sorted(expected2.columns[0].tolist())


CONDITIONS = [abi.PRED_EQUALS, abi.PRED_NOT_EQUALS, abi.PRED_LESS_THAN, abi.PRED_LESS_THAN_EQUALS,
              abi.PRED_GREATER_THAN, abi.PRED_GREATER_THAN_EQUALS, abi.PRED_BETWEEN_INCLUSIVE,
              abi.PRED_BETWEEN_LOWER_EXCLUSIVE, abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.PRED_BETWEEN_EXCLUSIVE,
              abi.PRED_IS_NULL, abi.PRED_IS_NOT_NULL]


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_oracle_matches_brute_force(np_type, with_nulls):
    rng = np.random.default_rng(7)
    n, chunk = 5000, 777
    values = rng.integers(-50, 50, n).astype(np_type)
    nulls = (rng.random(n) < 0.1) if with_nulls else None
    data_type = storage.TYPE_OF_NP[np.dtype(np_type)]
    for encoding in ENCODINGS:
        if encoding == abi.ENC_FRAME_OF_REFERENCE and np_type != np.int32:
            continue
        column = build_column(values, nulls, chunk, encoding, nullable=with_nulls)
        for condition in CONDITIONS:
            for value, value2 in ((-7, 12), (3, 3), (12, -7), (-1000, 1000), (49, 49)):
                result = oracle_scan(column, make_predicate(condition, data_type, value, value2, nullable=with_nulls),
                                     flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
                mask = brute_force_scan(values, nulls, condition, value, value2)
                expected = expected_result_from_mask(mask, chunk)
                np.testing.assert_array_equal(result.matches[:result.total], expected,
                                              err_msg=f"enc {encoding} cond {condition} values {value},{value2}")


def test_oracle_column_vs_column():
    rng = np.random.default_rng(11)
    n, chunk = 3000, 500
    left = rng.integers(0, 20, n).astype(np.int32)
    right_int = rng.integers(0, 20, n).astype(np.int32)
    right_float = rng.integers(0, 20, n).astype(np.float32) + np.float32(0.5) * (rng.random(n) < 0.5)
    lnull, rnull = rng.random(n) < 0.1, rng.random(n) < 0.1
    for right, rtype in ((right_int, None), (right_float, None)):
        for lenc in ENCODINGS:
            for renc in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
                lcol = build_column(left, lnull, chunk, lenc)
                rcol = build_column(right, rnull, chunk, renc)
                for condition in CONDITIONS[:6]:
                    result = oracle_scan_columns(lcol, rcol, condition)
                    ops = {abi.PRED_EQUALS: left == right, abi.PRED_NOT_EQUALS: left != right,
                           abi.PRED_LESS_THAN: left < right, abi.PRED_LESS_THAN_EQUALS: left <= right,
                           abi.PRED_GREATER_THAN: left > right, abi.PRED_GREATER_THAN_EQUALS: left >= right}
                    expected = expected_result_from_mask(ops[condition] & ~lnull & ~rnull, chunk)
                    np.testing.assert_array_equal(result.matches[:result.total], expected)


def test_string_date_twin_scans_like_the_int_column():
    """l_shipdate as DictionarySegment<pmr_string> of ISO dates (the reference's TPC-H schema, tpch_table_generator.cpp:46): the literal is
    resolved per chunk on the host (operators.string_predicate: lower_bound / upper_bound in the chunk's dictionary,
    column_vs_value_table_scan_impl.cpp:211-226, column_between_table_scan_impl.cpp:112-124) and the scan over value ids gives the
    PosLists, counts and early-out states of the same scan over the int twin."""
    from hyrise_amd import tpch
    from hyrise_amd.operators import string_predicate
    data = tpch.TpchData(scale_factor=0.05, seed=3)
    ints = storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY, 20_000)
    strings, dictionaries = tpch.string_date_column(ints)
    assert strings.data_type == abi.TYPE_STRING and dictionaries[0][0] < dictionaries[0][-1] and len(dictionaries[0][0]) == 10
    cases = [(abi.PRED_LESS_THAN, "1995-01-01", None, tpch.DAY_1995_01_01, None), (abi.PRED_LESS_THAN_EQUALS, "1998-09-02", None, tpch.DAY_1998_09_02, None),
             (abi.PRED_GREATER_THAN, "1995-06-17", None, tpch.CURRENT_DATE, None), (abi.PRED_GREATER_THAN_EQUALS, "1992-01-01", None, 0, None),
             (abi.PRED_EQUALS, "1995-06-17", None, tpch.CURRENT_DATE, None), (abi.PRED_NOT_EQUALS, "1995-06-17", None, tpch.CURRENT_DATE, None),
             (abi.PRED_EQUALS, "1995-06-17x", None, None, None),                                        # a literal that is in no dictionary
             (abi.PRED_BETWEEN_INCLUSIVE, "1994-01-01", "1994-12-31", tpch.DAY_1994_01_01, tpch.DAY_1995_01_01 - 1),
             (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, "1994-01-01", "1995-01-01", tpch.DAY_1994_01_01, tpch.DAY_1995_01_01),
             (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, "1993-12-31", "1994-12-31", tpch.DAY_1994_01_01 - 1, tpch.DAY_1995_01_01 - 1),
             (abi.PRED_BETWEEN_EXCLUSIVE, "1993-12-31", "1995-01-01", tpch.DAY_1994_01_01 - 1, tpch.DAY_1995_01_01)]
    for condition, a, b, ia, ib in cases:
        got = oracle_scan(strings, string_predicate(condition, dictionaries, a, b))
        if ia is None:
            assert got.total == 0
            continue
        want = oracle_scan(ints, make_predicate(condition, abi.TYPE_INT, ia, ib))
        assert got.total == want.total and got.matches[:got.total].tobytes() == want.matches[:want.total].tobytes(), condition
        np.testing.assert_array_equal(got.counts, want.counts)
        np.testing.assert_array_equal(got.chunk_state, want.chunk_state)


BETWEEN_TYPES = [np.int32, np.int64, np.float32, np.float64]


def between_column(a, nulls, encoding, nullable):
    """Chunk size 6, the two full chunks encoded, the third left as it was appended (table_scan_between_test.cpp:88-92).
    RunLength segments are expanded the way the residency cache expands them on upload (the oracle reads plain segments)."""
    if encoding != abi.ENC_RUN_LENGTH:
        return build_column(a, nulls, 6, [encoding, encoding], nullable=nullable)
    plain = build_column(a, nulls, 6, abi.ENC_UNENCODED, nullable=nullable)
    runs = [storage.encode_run_length(a[c * 6:c * 6 + 6], None if nulls is None else nulls[c * 6:c * 6 + 6]) for c in range(2)]
    return storage.expand_run_length(storage.HostColumn(runs + plain.segments[2:], plain.data_type))


@pytest.mark.parametrize("np_type", BETWEEN_TYPES, ids=lambda t: t.__name__)
@pytest.mark.parametrize("encoding", ENCODINGS + [abi.ENC_RUN_LENGTH], ids=ENC_IDS + ["RunLength"])
@pytest.mark.parametrize("sort_mode", KA.BETWEEN_SORT_MODES)
@pytest.mark.parametrize("nullable", [False, True], ids=["not_null", "nullable"])
def test_between_known_answers(np_type, encoding, sort_mode, nullable):
    """table_scan_between_test.cpp:194-243 (Inclusive / LowerExclusive / UpperExclusive / Exclusive) over its
    create_test_params() grid of numeric types x encodings x sort modes x nullability."""
    if encoding == abi.ENC_FRAME_OF_REFERENCE and np_type != np.int32:
        pytest.skip("encoding_supports_data_type(): FrameOfReference holds int only")
    a, nulls, b = KA.between_table(np_type, sort_mode, nullable)
    column = between_column(a, nulls, encoding, nullable)
    data_type = storage.TYPE_OF_NP[np.dtype(np_type)]
    cast = (lambda x: np_type(int(x))) if np.issubdtype(np_type, np.integer) else np_type   # static_cast<ColumnDataType>
    for condition, tests in KA.BETWEEN_TESTS.items():
        for lower, upper, expected in tests:
            p = make_predicate(condition, data_type, cast(lower), cast(upper), nullable=nullable)
            rows = np.array([c * 6 + o for c, o in result_rows(oracle_scan(column, p))], dtype=np.int64)
            assert sorted(b[rows].tolist()) == KA.between_expected(expected, sort_mode, nullable), \
                f"condition {condition} BETWEEN {lower} AND {upper}"


@pytest.mark.parametrize("sort_mode", KA.SORTED_SEGMENT_SORT_MODES)
@pytest.mark.parametrize("null_usage", KA.SORTED_SEGMENT_NULL_USAGES)
def test_sorted_segment_search_known_answers(sort_mode, null_usage):
    """table_scan_sorted_segment_search_test.cpp:106-214: the full scan emits what SortedSegmentSearch emits, in the same order."""
    values, nulls = KA.sorted_search_segment(sort_mode, null_usage)
    column = build_column(values, nulls, len(values), abi.ENC_UNENCODED, nullable=nulls is not None)
    for condition, value, value2, expected in KA.SORTED_SEGMENT_SEARCH_TESTS:
        result = oracle_scan(column, make_predicate(condition, abi.TYPE_INT, value, value2, nullable=nulls is not None))
        rows = [o for _, o in result_rows(result)]
        if null_usage == "OnlyNulls":
            assert rows == []
            continue
        assert not (nulls is not None and nulls[rows].any())
        want = expected if sort_mode.startswith("Ascending") else expected[::-1]
        assert values[rows].tolist() == want, f"condition {condition} value {value} / {value2}"

