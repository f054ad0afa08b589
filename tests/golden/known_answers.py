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
