"""hyrise_amd/join_keys.py: the ids string keys join as (CPU only; the hash itself is pinned against g++ in
tests/test_oracle_join.py)."""
import numpy as np

from hyrise_amd import abi, storage
from hyrise_amd.join_keys import HASH_BITS, StringJoinKeys, std_hash_bytes


def test_ids_are_unique_stable_and_carry_the_hash_in_their_low_bits():
    keys = StringJoinKeys()
    words = [b"", b"a", b"b", b"ab", b"Dampfschifffahrtsgesellschaft", "kapitän".encode()] + [b"w%d" % i for i in range(5000)]
    ids = [keys.key(w) for w in words]
    assert len(set(ids)) == len(words)
    assert ids == [keys.key(w) for w in words]                       # asking again does not renumber
    assert keys.key("ab") == keys.key(b"ab")                         # str is its UTF-8 bytes
    mask = (1 << HASH_BITS) - 1
    for w, i in zip(words, ids):
        assert i > 0 and i & mask == std_hash_bytes(w) & mask        # radix partition (<= 8 bits) and Bloom index (20 bits)
    assert max(ids) < 1 << 62


def test_column_view_keeps_the_attribute_vectors_and_maps_null_value_ids():
    keys = StringJoinKeys()
    values = [b"x", b"y", b"x", b"z", b"y", b"q", b"x"]
    nulls = np.array([0, 0, 0, 1, 0, 0, 0], dtype=bool)
    segments, dictionaries = [], []
    for begin in (0, 4):
        segment, dictionary = storage.encode_string_dictionary(values[begin:begin + 4], nulls[begin:begin + 4])
        segments.append(segment)
        dictionaries.append(dictionary)
    column = keys.column(segments, dictionaries)
    assert column.data_type == abi.TYPE_LONG and column.n_chunks == 2
    for view, segment, dictionary in zip(column.segments, segments, dictionaries):
        assert view.encoding == abi.ENC_DICTIONARY and view.data is segment.data and view.width == segment.width
        assert view.aux.dtype == np.int64 and list(view.aux) == [keys.key(entry) for entry in dictionary]
        assert view.aux_size == len(dictionary)                      # the NULL value id stays the dictionary size
    assert int(segments[0].data[3]) == len(dictionaries[0])          # row 3 is NULL
    # the same string in two chunks gets the same id although its value ids differ
    first = {entry: int(i) for entry, i in zip(dictionaries[0], column.segments[0].aux)}
    second = {entry: int(i) for entry, i in zip(dictionaries[1], column.segments[1].aux)}
    assert first[b"x"] == second[b"x"] and first[b"y"] == second[b"y"]
