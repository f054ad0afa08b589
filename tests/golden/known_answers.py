"""Expected values that the reference keeps INLINE in its gtest sources (restated with citations; the .tbl fixtures
they run on are under tests/golden/tbl/, copied by make_golden.py).  All paths: /root/reference/src/test/lib/operators/.
"""
from hyrise_amd import abi

# table_scan_test.cpp:407-431  ScanOnCompressedSegments: scan column a of int_int_shuffled(.tbl, chunk 7) and
# int_int_shuffled_2(.tbl, chunk 5) with literal 6; expected values of column b (as a multiset).
SCAN_ON_COMPRESSED_SEGMENTS = {
    abi.PRED_EQUALS: [106, 106],
    abi.PRED_NOT_EQUALS: [100, 102, 104, 108, 110, 112, 100, 102, 104, 108, 110, 112],
    abi.PRED_LESS_THAN: [100, 102, 104, 100, 102, 104],
    abi.PRED_LESS_THAN_EQUALS: [100, 102, 104, 106, 100, 102, 104, 106],
    abi.PRED_GREATER_THAN: [108, 110, 112, 108, 110, 112],
    abi.PRED_GREATER_THAN_EQUALS: [106, 108, 110, 112, 106, 108, 110, 112],
    abi.PRED_IS_NULL: [],
    abi.PRED_IS_NOT_NULL: [100, 102, 104, 106, 108, 110, 112, 100, 102, 104, 106, 108, 110, 112],
}

# table_scan_test.cpp:433-463  ScanOnReferencedCompressedSegments: first `b < 108` (all of those rows), then on that
# reference table `a <op> 4`; expected column b.
SCAN_ON_REFERENCED_COMPRESSED_SEGMENTS = {
    abi.PRED_EQUALS: [104, 104],
    abi.PRED_NOT_EQUALS: [100, 102, 106, 100, 102, 106],
    abi.PRED_LESS_THAN: [100, 102, 100, 102],
    abi.PRED_LESS_THAN_EQUALS: [100, 102, 104, 100, 102, 104],
    abi.PRED_GREATER_THAN: [106, 106],
    abi.PRED_GREATER_THAN_EQUALS: [104, 106, 104, 106],
    abi.PRED_IS_NULL: [],
    abi.PRED_IS_NOT_NULL: [100, 102, 104, 106, 100, 102, 104, 106],
}

# table_scan_test.cpp:126-152 + 465-484  ScanWeirdPosList: one reference chunk over int_int_shuffled_2 (chunk 5) with
# this multi-chunk pos list; scan a <op> 10; expected column b.
WEIRD_POS_LIST = [(2, 0), (1, 1), (1, 3), (0, 2), (2, 2), (0, 0), (0, 4)]
SCAN_WEIRD_POS_LIST = {
    abi.PRED_EQUALS: [110, 110],
    abi.PRED_NOT_EQUALS: [100, 102, 106, 108, 112],
    abi.PRED_LESS_THAN: [100, 102, 106, 108],
    abi.PRED_LESS_THAN_EQUALS: [100, 102, 106, 108, 110, 110],
    abi.PRED_GREATER_THAN: [112],
    abi.PRED_GREATER_THAN_EQUALS: [110, 110, 112],
    abi.PRED_IS_NULL: [],
    abi.PRED_IS_NOT_NULL: [100, 102, 106, 108, 110, 110, 112],
}

ALL_ROWS_SHUFFLED = [100, 102, 104, 106, 108, 110, 112, 100, 102, 104, 106, 108, 110, 112]
# table_scan_test.cpp:486-509  literal 30 > every dictionary value
SCAN_VALUE_GREATER_THAN_MAX = {
    abi.PRED_EQUALS: [], abi.PRED_NOT_EQUALS: ALL_ROWS_SHUFFLED, abi.PRED_LESS_THAN: ALL_ROWS_SHUFFLED,
    abi.PRED_LESS_THAN_EQUALS: ALL_ROWS_SHUFFLED, abi.PRED_GREATER_THAN: [], abi.PRED_GREATER_THAN_EQUALS: [],
}
# table_scan_test.cpp:511-534  literal -10 < every dictionary value
SCAN_VALUE_LESS_THAN_MIN = {
    abi.PRED_EQUALS: [], abi.PRED_NOT_EQUALS: ALL_ROWS_SHUFFLED, abi.PRED_LESS_THAN: [],
    abi.PRED_LESS_THAN_EQUALS: [], abi.PRED_GREATER_THAN: ALL_ROWS_SHUFFLED,
    abi.PRED_GREATER_THAN_EQUALS: ALL_ROWS_SHUFFLED,
}

# table_scan_test.cpp:622-652  ScanOnWideDictionarySegment: table of 0..n (one chunk), `a > literal` row counts.
WIDE_DICTIONARY = [((1 << 8) + 1, 200, 57), ((1 << 16) + 1, 65500, 37)]

# table_scan_test.cpp:661-684  int_int_w_null_8_rows.tbl (chunk 4): scan column b IS [NOT] NULL, expected column a
# (None == NULL_VALUE).
SCAN_FOR_NULL_VALUES = {
    abi.PRED_IS_NULL: [12, 123],
    abi.PRED_IS_NOT_NULL: [12345, None, 1234, 12345, 12, 1234],
}

# table_scan_test.cpp:296-302  SingleScan: int_float.tbl `a >= 1234` == int_float_filtered2.tbl
# table_scan_test.cpp:330-342 (DoubleScan): `a >= 1234` then `b < 457.9` == int_float_filtered.tbl

# table_scan_between_test.cpp:194-243  (lower bound, upper bound, expected values of column b).  The table (set-up at
# :40-96): column a = cast<ColumnType>(10.25 + 2 i) for i = 0..10 (30.25 - 2 i when sorted descending), column b = row
# index, chunk size 6 with the first two chunks encoded and the last left unencoded; nullable + unsorted: every i with
# i % 3 == 2 is NULL; nullable + sorted: three NULL rows in front.  Both bounds are cast to the column type (:146-147),
# so an int column sees BETWEEN 12 AND 16 for (12.25, 16.75) -- the lists hold for int, long, float and double alike.
BETWEEN_TESTS = {
    abi.PRED_BETWEEN_INCLUSIVE: [
        (12.25, 16.25, [1, 2, 3]), (12.0, 16.25, [1, 2, 3]), (12.25, 16.75, [1, 2, 3]), (12.0, 16.75, [1, 2, 3]),
        (0.0, 16.75, [0, 1, 2, 3]), (16.0, 50.75, [3, 4, 5, 6, 7, 8, 9, 10]), (13.0, 16.25, [2, 3]),
        (12.25, 15.0, [1, 2]), (0.25, 50.75, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]), (0.25, 0.75, []),
    ],
    abi.PRED_BETWEEN_LOWER_EXCLUSIVE: [(11.0, 16.25, [1, 2, 3]), (12.25, 16.25, [2, 3]), (13.0, 16.25, [2, 3])],
    abi.PRED_BETWEEN_UPPER_EXCLUSIVE: [(12.25, 17.0, [1, 2, 3]), (12.25, 16.25, [1, 2]), (12.25, 15.0, [1, 2])],
    abi.PRED_BETWEEN_EXCLUSIVE: [
        (12.25, 16.25, [2]), (11.0, 16.25, [1, 2]), (12.25, 17.0, [2, 3]), (11.0, 17.0, [1, 2, 3]), (13.0, 16.25, [2]),
        (12.25, 15.0, [2]), (13.0, 15.0, [2]),
    ],
}
BETWEEN_SORT_MODES = ["unsorted", "ascending", "descending"]


def between_table(np_type, sort_mode, nullable):
    """(values of column a, NULL mask or None, values of column b): table_scan_between_test.cpp:40-78."""
    import numpy as np
    leading_nulls = 3 if (nullable and sort_mode != "unsorted") else 0
    a, nulls, b = [0] * leading_nulls, [True] * leading_nulls, list(range(leading_nulls))
    for i in range(11):
        value = 30.25 - 2.0 * i if sort_mode == "descending" else 10.25 + 2.0 * i
        null = nullable and sort_mode == "unsorted" and i % 3 == 2
        a.append(0 if null else np_type(int(value)) if np.issubdtype(np_type, np.integer) else np_type(value))
        nulls.append(null)
        b.append(i + leading_nulls)
    return np.array(a, dtype=np_type), (np.array(nulls, dtype=bool) if nullable else None), np.array(b, dtype=np.int32)


def between_expected(expected, sort_mode, nullable):
    """table_scan_between_test.cpp:167-191: the index lists above moved to where the rows are in this variant."""
    leading_nulls = 3 if (nullable and sort_mode != "unsorted") else 0
    if sort_mode == "descending":
        return sorted(10 + leading_nulls - x for x in expected)
    if sort_mode == "ascending":
        return [x + leading_nulls for x in expected]
    return [x for x in expected if not (nullable and x % 3 == 2)]

# table_scan_sorted_segment_search_test.cpp:106-171  (condition, value, second value, expected VALUES in position order on the ascending
# segment 0, 0, 1, 1, 2, 2, 3, 3, 4, 4; descending segments hold 4, 4, ..., 0, 0 and expect the reversed list, :58-60).  The segment is a
# ValueSegment<int32_t> with three NULLs in front (NullsFirst) or behind (NullsLast) when nullable, or three NULLs only (:64-90).
# SortedSegmentSearch is the reference's shortcut for sorted chunks; a scan that ignores the sort flags must emit the same positions in
# the same order -- which is what these vectors pin for the full scan here.
SORTED_SEGMENT_SEARCH_TESTS = [
    (abi.PRED_EQUALS, 2, None, [2, 2]),
    (abi.PRED_NOT_EQUALS, 2, None, [0, 0, 1, 1, 3, 3, 4, 4]), (abi.PRED_NOT_EQUALS, 4, None, [0, 0, 1, 1, 2, 2, 3, 3]),
    (abi.PRED_NOT_EQUALS, 0, None, [1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_NOT_EQUALS, 5, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_NOT_EQUALS, -1, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_LESS_THAN, -1, None, []), (abi.PRED_LESS_THAN, 0, None, []), (abi.PRED_LESS_THAN, 2, None, [0, 0, 1, 1]),
    (abi.PRED_LESS_THAN, 5, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_LESS_THAN, 4, None, [0, 0, 1, 1, 2, 2, 3, 3]),
    (abi.PRED_LESS_THAN_EQUALS, -1, None, []), (abi.PRED_LESS_THAN_EQUALS, 0, None, [0, 0]), (abi.PRED_LESS_THAN_EQUALS, 2, None, [0, 0, 1, 1, 2, 2]),
    (abi.PRED_LESS_THAN_EQUALS, 5, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_LESS_THAN_EQUALS, 4, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_GREATER_THAN, -1, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_GREATER_THAN, 0, None, [1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_GREATER_THAN, 2, None, [3, 3, 4, 4]), (abi.PRED_GREATER_THAN, 5, None, []), (abi.PRED_GREATER_THAN, 4, None, []),
    (abi.PRED_GREATER_THAN_EQUALS, -1, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_GREATER_THAN_EQUALS, 0, None, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_GREATER_THAN_EQUALS, 2, None, [2, 2, 3, 3, 4, 4]), (abi.PRED_GREATER_THAN_EQUALS, 5, None, []), (abi.PRED_GREATER_THAN_EQUALS, 4, None, [4, 4]),
    (abi.PRED_BETWEEN_INCLUSIVE, -1, 5, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_BETWEEN_INCLUSIVE, 0, 4, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_INCLUSIVE, 0, 3, [0, 0, 1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_INCLUSIVE, 2, 4, [2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_INCLUSIVE, 1, 3, [1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_INCLUSIVE, 5, 10, []), (abi.PRED_BETWEEN_INCLUSIVE, -5, -1, []),
    (abi.PRED_BETWEEN_EXCLUSIVE, -2, 6, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_BETWEEN_EXCLUSIVE, -1, 5, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_EXCLUSIVE, -1, 4, [0, 0, 1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_EXCLUSIVE, 1, 5, [2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_EXCLUSIVE, 0, 4, [1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_EXCLUSIVE, 4, 10, []), (abi.PRED_BETWEEN_EXCLUSIVE, -5, 0, []),
    (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, -2, 4, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, -1, 4, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, -1, 3, [0, 0, 1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, 1, 4, [2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, 0, 3, [1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, 4, 10, []), (abi.PRED_BETWEEN_LOWER_EXCLUSIVE, -5, -1, []),
    (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, -1, 6, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]), (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 0, 5, [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 0, 4, [0, 0, 1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 2, 5, [2, 2, 3, 3, 4, 4]),
    (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 1, 4, [1, 1, 2, 2, 3, 3]), (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 5, 10, []), (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, -5, 0, []),
]
SORTED_SEGMENT_SORT_MODES = ["AscendingNullsFirst", "DescendingNullsFirst", "AscendingNullsLast", "DescendingNullsLast"]
SORTED_SEGMENT_NULL_USAGES = ["WithoutNulls", "WithNulls", "OnlyNulls"]


def sorted_search_segment(sort_mode, null_usage):
    """(values, NULL mask or None): table_scan_sorted_segment_search_test.cpp:62-90."""
    import numpy as np
    ascending, nulls_last = sort_mode.startswith("Ascending"), sort_mode.endswith("NullsLast")
    nullable, only_nulls = null_usage != "WithoutNulls", null_usage == "OnlyNulls"
    values, nulls = [], []
    if (nullable and not nulls_last) or only_nulls:
        values += [0, 0, 0]
        nulls += [True] * 3
    if not only_nulls:
        for row in range(5):
            values += [row if ascending else 4 - row] * 2
            nulls += [False, False]
        if nullable and nulls_last:
            values += [0, 0, 0]
            nulls += [True] * 3
    return np.array(values, dtype=np.int32), (np.array(nulls, dtype=bool) if nullable else None)
