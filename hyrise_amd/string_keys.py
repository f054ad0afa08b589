"""String columns for AggregateHash, on the host side of the boundary.

The device groups and aggregates 64-bit integers; a `pmr_string` column reaches it as integers the adapter derives per
DISTINCT string of a chunk's dictionary (never per row), with the attribute vector passed through unchanged:

  GROUP BY column   the string's AggregateKeyEntry -- the very number AggregateHash itself puts into its AggregateKey
                    (aggregate_hash.cpp:852-914): 1 for "", 2 + byte, 258 + 2 bytes, 65 794 + 3 bytes, 16 843 010 + 4 bytes
                    (little endian), and for strings of five or more characters ids handed out from 5 000 000 000 in order
                    of first appearance.  Equal numbers <=> equal strings, which is all GROUP BY needs (the group's string is
                    read back through its representative RowID, write_groupby_output).
  aggregate column  (MIN / MAX / ANY / COUNT / COUNT DISTINCT over strings, window_function_traits.hpp:15-59) the string's
                    rank among the column's distinct strings in byte order: order preserving, so MIN / MAX of the ranks is the
                    rank of the MIN / MAX string; `strings_of()` maps result ranks back.

TPC-H Q1 groups by l_returnflag / l_linestatus, one-character strings: their key is 2 + the character code."""
import numpy as np

from . import abi
from .storage import HostColumn, HostSegment, fixed_width

LONG_STRING_IDS_FROM = 5_000_000_000   # aggregate_hash.cpp:824-826


def _bytes(value):
    return value if isinstance(value, bytes) else str(value).encode("utf-8")


class AggregateKeyNames:
    """AggregateKeyEntry of strings; one instance per GROUP BY column of one operator run (the map ids of long strings
    follow the order of first appearance, like the reference's id_map)."""

    def __init__(self, shared_long_strings=None):
        # shared_long_strings: the distinct strings of five or more bytes of the WHOLE column, in an order every rank agrees on
        # (distributed.shared_long_strings): their ids are then the same on every rank, which a cross-rank merge by key value needs.
        # Without it the ids are names inside ONE process only (`shared` stays False and the sharded operators refuse long strings).
        self._long = {}
        self.shared = shared_long_strings is not None
        for data in shared_long_strings or ():
            self._long.setdefault(_bytes(data), LONG_STRING_IDS_FROM + len(self._long))

    @property
    def has_long_strings(self):
        return bool(self._long)

    def name(self, value):
        data = _bytes(value)
        if len(data) == 0:
            return 1
        if len(data) < 5:
            base = (2, 258, 65_794, 16_843_010)[len(data) - 1]
            return base + int.from_bytes(data, "little")
        found = self._long.get(data)
        if found is None:
            found = LONG_STRING_IDS_FROM + len(self._long)
            self._long[data] = found
        return found

    def dictionary_column(self, segments, dictionaries):
        """segments / dictionaries per chunk as storage.encode_string_dictionary returns them -> HostColumn of
        DictionarySegment<int64> views: same attribute vectors, dictionaries of names (chunk order = first appearance)."""
        out = []
        for segment, dictionary in zip(segments, dictionaries):
            names = np.array([self.name(entry) for entry in dictionary], dtype=np.int64)
            out.append(HostSegment(abi.ENC_DICTIONARY, abi.TYPE_LONG, segment.size, segment.width, segment.data, aux=names, aux_size=len(names)))
        return HostColumn(out, abi.TYPE_LONG)

    def value_column(self, values, nulls, chunk_size):
        """An unencoded string column, row by row (ValueSegment<pmr_string>: the adapter would rather dictionary-encode it)."""
        names = np.array([0 if (nulls is not None and nulls[i]) else self.name(v) for i, v in enumerate(values)], dtype=np.int64)
        from .storage import make_column
        return make_column(names, nulls, abi.ENC_UNENCODED, chunk_size)


class StringRanks:
    """Order-preserving stand-ins for the strings of one column: rank among its distinct strings (byte order)."""

    def __init__(self, values, nulls=None):
        present = {_bytes(v) for i, v in enumerate(values) if nulls is None or not nulls[i]}
        self.sorted = sorted(present)
        self._rank = {s: i for i, s in enumerate(self.sorted)}

    def rank(self, value):
        return self._rank[_bytes(value)]

    def strings_of(self, ranks):
        return [None if r is None else self.sorted[int(r)].decode("utf-8") for r in ranks]

    def dictionary_column(self, segments, dictionaries):
        out = []
        for segment, dictionary in zip(segments, dictionaries):
            ranks = np.array([self._rank[_bytes(entry)] for entry in dictionary], dtype=np.int64)   # ascending, like the dictionary
            out.append(HostSegment(abi.ENC_DICTIONARY, abi.TYPE_LONG, segment.size, segment.width, segment.data, aux=ranks, aux_size=len(ranks)))
        return HostColumn(out, abi.TYPE_LONG)

    def value_column(self, values, nulls, chunk_size):
        ranks = np.array([0 if (nulls is not None and nulls[i]) else self.rank(v) for i, v in enumerate(values)], dtype=np.int64)
        from .storage import make_column
        return make_column(ranks, nulls, abi.ENC_UNENCODED, chunk_size)


def encode_string_column(values, nulls, chunk_size):
    """DictionarySegment<pmr_string> per chunk -> (segments, dictionaries) for the classes above."""
    from .storage import encode_string_dictionary
    segments, dictionaries = [], []
    for begin in range(0, len(values), chunk_size):
        end = min(len(values), begin + chunk_size)
        segment, dictionary = encode_string_dictionary(values[begin:end], None if nulls is None else nulls[begin:end])
        segments.append(segment)
        dictionaries.append(dictionary)
    return segments, dictionaries
