"""The reference's LIKE tests on compressed string columns (src/test/lib/operators/table_scan_string_test.cpp) as data:
(line in that file, condition, pattern, expected .tbl or None for "no rows").  Input: int_string_like.tbl at chunk size 5
(:45), column b dictionary-encoded; the special-character cases (:193-219) use int_string_like_special_chars.tbl."""
import numpy as np

from hyrise_amd import abi, storage
from hyrise_amd.like import dictionary_matches
from hyrise_amd.operators import make_predicate
from support import load_tbl

STRING_TABLE_CASES = [
    (116, abi.PRED_LIKE, "%", "int_string_like_without_null.tbl"),
    (125, abi.PRED_LIKE, "%D%_m_f%", "int_string_like_starting.tbl"),
    (151, abi.PRED_LIKE, "Dampf%", "int_string_like_starting.tbl"),
    (178, abi.PRED_LIKE, "%gesellschaft", "int_string_like_ending.tbl"),
    (224, abi.PRED_LIKE, "Schiff%schaft", "int_string_like_containing_wildcard.tbl"),
    (242, abi.PRED_LIKE, "%schifffahrtsgesellschaft%", "int_string_like_containing.tbl"),
    (265, abi.PRED_LIKE, "%not_there%", None),
    (288, abi.PRED_NOT_LIKE, "%", None),
    (306, abi.PRED_NOT_LIKE, "%foo%", "int_string_like_without_null.tbl"),
    (324, abi.PRED_NOT_LIKE, "D_m_f%", "int_string_like_not_starting.tbl"),
    # case-insensitive variants of the same expectations (like_matcher.hpp:74-85 lower-cases both sides)
    (151, abi.PRED_LIKE_INSENSITIVE, "dAMPF%", "int_string_like_starting.tbl"),
    (324, abi.PRED_NOT_LIKE_INSENSITIVE, "d_M_f%", "int_string_like_not_starting.tbl"),
]
SPECIAL_CHARS_CASES = [
    (201, abi.PRED_LIKE, "%2^2%", "int_string_like_special_chars_1.tbl"),
    (205, abi.PRED_LIKE, "%$%$%", "int_string_like_special_chars_1.tbl"),
    (211, abi.PRED_LIKE, "%(%)%", "int_string_like_special_chars_2.tbl"),
    (215, abi.PRED_LIKE, "%la\\.^$+?)({}.*__bl%", "int_string_like_special_chars_3.tbl"),
]


class StringTable:
    """A .tbl fixture with its string column dictionary-encoded chunk by chunk."""

    def __init__(self, name, chunk_size, string_column=1):
        self.tbl = load_tbl(name)
        self.chunk_size = chunk_size
        values, nulls = self.tbl.columns[string_column], self.tbl.nulls[string_column]
        segments, self.dictionaries = [], []
        for begin in range(0, len(values), chunk_size):
            segment, dictionary = storage.encode_string_dictionary(values[begin:begin + chunk_size], nulls[begin:begin + chunk_size])
            segments.append(segment)
            self.dictionaries.append(dictionary)
        self.column = storage.HostColumn(segments, abi.TYPE_STRING)
        self.string_column = string_column

    def predicate(self, condition, pattern):
        return make_predicate(condition, abi.TYPE_STRING, nullable=self.tbl.nullable[self.string_column],
                              dictionary_matches=dictionary_matches(self.dictionaries, pattern, condition))

    def rows_of(self, row_ids):
        """(a, b) tuples of the data table's rows `row_ids` [(chunk, offset)...], sorted (EXPECT_TABLE_EQ_UNORDERED)."""
        out = []
        for chunk, offset in row_ids:
            r = int(chunk) * self.chunk_size + int(offset)
            out.append(tuple(None if self.tbl.nulls[c][r] else (self.tbl.columns[c][r].item() if hasattr(self.tbl.columns[c][r], "item") else self.tbl.columns[c][r])
                             for c in range(len(self.tbl.columns))))
        return sorted(out, key=repr)


def expected_rows(name):
    if name is None:
        return []
    t = load_tbl(name)
    n = len(t.columns[0])
    rows = [tuple(None if t.nulls[c][r] else (t.columns[c][r].item() if hasattr(t.columns[c][r], "item") else t.columns[c][r])
                  for c in range(len(t.columns))) for r in range(n)]
    return sorted(rows, key=repr)
