"""String join keys for JoinHash, on the host side of the boundary.

JoinHash hashes `pmr_string` keys with std::hash (join_hash_steps.hpp:282,352,573: Bloom index = hash % 2^20, radix
partition = hash & mask) and compares them for equality in the hash table.  The device joins 64-bit integers whose radix
partition and Bloom index are their LOW BITS, so the adapter gives every distinct string the id

    unique number << 20  |  std::hash(string) & 0xFFFFF

-- equal ids <=> equal strings, `id & mask` == `hash & mask` for every radix_bits <= 8, `id % 2^20` == the Bloom index --
and presents a DictionarySegment<pmr_string> as a DictionarySegment<int64> with the same attribute vector and a
dictionary of ids (the work is per DISTINCT string of a chunk, not per row).  The join over those columns is the
reference's join over the strings: same partitions, same probe-row order, same build-side insertion order.

std::hash<pmr_string> is libstdc++'s _Hash_bytes (libsupc++/hash_bytes.cc: the 64-bit Murmur-style function, seed
0xc70f6907) over the string's bytes; tests/test_oracle_join.py pins this restatement against the g++ installed here."""
import numpy as np

from . import abi
from .storage import HostColumn, HostSegment

_MASK = (1 << 64) - 1
_MUL = ((0xc6a4a793 << 32) + 0x5bd1e995) & _MASK
_SEED = 0xc70f6907
HASH_BITS = 20   # log2(BLOOM_FILTER_SIZE), join_hash_steps.hpp:252


def _shift_mix(v):
    return v ^ (v >> 47)


def std_hash_bytes(data):
    """libstdc++'s std::hash<std::string>{}(data) on a 64-bit target."""
    data = bytes(data)
    length = len(data)
    aligned = length & ~7
    value = (_SEED ^ (length * _MUL)) & _MASK
    for p in range(0, aligned, 8):
        word = int.from_bytes(data[p:p + 8], "little")
        value ^= (_shift_mix((word * _MUL) & _MASK) * _MUL) & _MASK
        value = (value * _MUL) & _MASK
    if length & 7:
        value ^= int.from_bytes(data[aligned:], "little")
        value = (value * _MUL) & _MASK
    value = (_shift_mix(value) * _MUL) & _MASK
    return _shift_mix(value)


class StringJoinKeys:
    """The id of every distinct string seen so far (one registry for both sides of a join -- or for a whole database)."""

    def __init__(self):
        self._ids = {}

    def key(self, value):
        value = value if isinstance(value, bytes) else str(value).encode("utf-8")
        found = self._ids.get(value)
        if found is None:
            found = ((len(self._ids) + 1) << HASH_BITS) | (std_hash_bytes(value) & ((1 << HASH_BITS) - 1))
            self._ids[value] = found
        return found

    def column(self, segments, dictionaries):
        """segments / dictionaries: per chunk, what storage.encode_string_dictionary returned.  -> HostColumn (int64)."""
        out = []
        for segment, dictionary in zip(segments, dictionaries):
            assert segment.encoding == abi.ENC_DICTIONARY and segment.aux_size == len(dictionary)
            ids = np.array([self.key(entry) for entry in dictionary], dtype=np.int64)
            out.append(HostSegment(abi.ENC_DICTIONARY, abi.TYPE_LONG, segment.size, segment.width, segment.data, aux=ids, aux_size=len(ids)))
        return HostColumn(out, abi.TYPE_LONG)
