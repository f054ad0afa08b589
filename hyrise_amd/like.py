"""Host side of the LIKE family: evaluates the pattern over a chunk's (small) string dictionary and hands the device a
bitmap over value ids -- the split ColumnLikeTableScanImpl::_scan_dictionary_segment makes
(src/lib/operators/table_scan/column_like_table_scan_impl.cpp:74-140). Pattern semantics follow LikeMatcher
(src/lib/expression/evaluation/like_matcher.cpp:32-55): `%` any run of bytes, `_` exactly one byte (bytes, not code points:
the reference matches pmr_string chars), everything else literal; the *Insensitive conditions lower-case both sides
(like_matcher.hpp:74-85)."""
import re

import numpy as np

from . import abi

_LIKE_CONDITIONS = (abi.PRED_LIKE, abi.PRED_NOT_LIKE, abi.PRED_LIKE_INSENSITIVE, abi.PRED_NOT_LIKE_INSENSITIVE)


def _as_bytes(value):
    return value if isinstance(value, bytes) else str(value).encode("utf-8")


def like_to_regex(pattern):
    """SQL LIKE pattern -> anchored bytes regex."""
    out = [b"^"]
    for byte in _as_bytes(pattern):
        ch = bytes([byte])
        out.append(b".*" if ch == b"%" else b"." if ch == b"_" else re.escape(ch))
    out.append(b"$")
    return re.compile(b"".join(out), re.DOTALL)


class LikeMatcher:
    def __init__(self, pattern, condition=abi.PRED_LIKE):
        if condition not in _LIKE_CONDITIONS:
            raise ValueError("Expected PredicateCondition (Not)Like or (Not)LikeInsensitive.")
        self.insensitive = condition in (abi.PRED_LIKE_INSENSITIVE, abi.PRED_NOT_LIKE_INSENSITIVE)
        self.negated = condition in (abi.PRED_NOT_LIKE, abi.PRED_NOT_LIKE_INSENSITIVE)
        pattern = _as_bytes(pattern)
        self.regex = like_to_regex(pattern.lower() if self.insensitive else pattern)

    def __call__(self, value):
        value = _as_bytes(value)
        if self.insensitive:
            value = value.lower()
        return (self.regex.match(value) is not None) != self.negated


def dictionary_matches(dictionaries, pattern, condition=abi.PRED_LIKE):
    """One bool array per data chunk over that chunk's dictionary (_find_matches_in_dictionary, :142-159)."""
    matcher = LikeMatcher(pattern, condition)
    return [np.fromiter((matcher(entry) for entry in dictionary), dtype=bool, count=len(dictionary)) for dictionary in dictionaries]
