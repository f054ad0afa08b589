"""LIKE / NOT LIKE on dictionary-encoded string columns: the oracle's restatement of
ColumnLikeTableScanImpl::_scan_dictionary_segment (column_like_table_scan_impl.cpp:74-140) plus the host-side LikeMatcher
(hyrise_amd/like.py), pinned against the reference's own expected tables (table_scan_string_test.cpp)."""
import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.like import LikeMatcher
from hyrise_amd.operators import make_predicate
from like_cases import SPECIAL_CHARS_CASES, STRING_TABLE_CASES, StringTable, expected_rows
from support import build_column, load_tbl, oracle_scan, result_rows


@pytest.mark.parametrize("line,condition,pattern,expected", STRING_TABLE_CASES)
def test_like_on_dictionary_segments(line, condition, pattern, expected):
    table = StringTable("int_string_like.tbl", 5)
    got = oracle_scan(table.column, table.predicate(condition, pattern))
    assert table.rows_of(result_rows(got)) == expected_rows(expected), f"table_scan_string_test.cpp:{line}"
    if expected is None:   # `%not_there%` / NOT LIKE `%`: every chunk is an early out (:114-118)
        assert all(state == abi.CHUNK_NONE_MATCH for state in got.chunk_state[:table.column.n_chunks])


@pytest.mark.parametrize("line,condition,pattern,expected", SPECIAL_CHARS_CASES)
def test_like_special_characters(line, condition, pattern, expected):
    table = StringTable("int_string_like_special_chars.tbl", 2)
    got = oracle_scan(table.column, table.predicate(condition, pattern))
    assert table.rows_of(result_rows(got)) == expected_rows(expected), f"table_scan_string_test.cpp:{line}"


@pytest.mark.parametrize("line,condition,pattern,expected", [c for c in STRING_TABLE_CASES if c[0] in (151, 178, 242, 265)])
def test_like_on_referenced_dictionary_segments(line, condition, pattern, expected):
    """ScanLike*OnReferencedDictSegment (:156-163,183-190,247-254,269-275): a > 0 first, then LIKE on the reference table."""
    table = StringTable("int_string_like.tbl", 5)
    a = build_column(table.tbl.columns[0], None, 5, abi.ENC_UNENCODED)
    first = oracle_scan(a, make_predicate(abi.PRED_GREATER_THAN, abi.TYPE_INT, 0), flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
    pos_lists = [first.pos_list(c).copy() for c in range(a.n_chunks)]
    referencing = storage.make_reference_column(table.column, pos_lists, list(range(a.n_chunks)))
    got = oracle_scan(referencing, table.predicate(condition, pattern))
    data_rows = [tuple(pos_lists[chunk][offset]) for chunk, offset in result_rows(got)]
    assert table.rows_of(data_rows) == expected_rows(expected), f"table_scan_string_test.cpp:{line}"


def test_like_matcher_semantics():
    """LikeMatcher (like_matcher.cpp:32-55): % and _ are the only wildcards, regex metacharacters are literals, `_` is
    one *byte* (the reference matches chars of a pmr_string)."""
    assert LikeMatcher("a.c")("a.c") and not LikeMatcher("a.c")("abc")
    assert LikeMatcher("a_c")("abc") and not LikeMatcher("a_c")("ac")
    assert LikeMatcher("%")("") and LikeMatcher("%%")("anything")
    assert LikeMatcher("_")("x") and not LikeMatcher("_")("ä")   # two bytes in UTF-8
    assert LikeMatcher("__")("ä")
    assert LikeMatcher("a%b", abi.PRED_NOT_LIKE)("xab") and not LikeMatcher("a%b", abi.PRED_NOT_LIKE)("a..b")
    assert LikeMatcher("HeLLo%", abi.PRED_LIKE_INSENSITIVE)("hello world")
    assert LikeMatcher("line\nbreak%")("line\nbreak and more\nlines")
    with pytest.raises(ValueError):
        LikeMatcher("x", abi.PRED_EQUALS)


def test_like_needs_dictionary_bitmaps():
    """Without the host's bitmaps (or on a numeric column) the oracle refuses, like ColumnLikeTableScanImpl's
    'LIKE operator only applicable on string columns' (:32-36)."""
    from support import OracleCol, oracle
    import ctypes as C
    from hyrise_amd.operators import HostScanResult
    ints = build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_DICTIONARY)
    col = OracleCol(ints)
    result = HostScanResult(ints.n_chunks, ints.rows)
    p = make_predicate(abi.PRED_LIKE, abi.TYPE_STRING)
    assert oracle().hyo_table_scan(C.byref(col.c), C.byref(p), C.byref(result.c), 1) != 0
