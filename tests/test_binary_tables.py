"""Hyrise `.bin` tables: the reader/writer of hyrise_amd/binary.py against files written
by Hyrise itself (tests/golden/bin, the byte-for-byte references of the reference's binary_writer_test.cpp), and -- the
point of the exercise -- this package's ENCODERS against real Hyrise output: re-encoding the decoded values must
reproduce the reference's DictionarySegment / FrameOfReferenceSegment bytes exactly."""
import glob
import os

import numpy as np
import pytest

from hyrise_amd import abi, binary, storage
from support import GOLDEN

BIN = os.path.join(os.path.dirname(GOLDEN), "bin")
FILES = sorted(p for p in glob.glob(os.path.join(BIN, "**", "*.bin"), recursive=True))
UNSUPPORTED = ("LZ4.bin",)


def supported(path):
    try:
        binary.read_table(path)
        return True
    except binary.UnsupportedSegment:
        return False


def test_fixtures_present():
    assert len(FILES) >= 40


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, BIN) for p in FILES])
def test_read_write_round_trip(path, tmp_path):
    if path.endswith(UNSUPPORTED) and not supported(path):
        with pytest.raises(binary.UnsupportedSegment):
            binary.read_table(path)
        return
    table = binary.read_table(path)
    assert binary.write_table(str(tmp_path / "out.bin"), table) == open(path, "rb").read()


def numeric_columns(table):
    return [c for c, t in enumerate(table.types) if t != abi.TYPE_STRING]


@pytest.mark.parametrize("directory", ["AllTypesAllNullValues", "AllTypesMixColumn", "AllTypesNullValues", "AllTypesSegmentSorted", "AllTypesSegmentUnsorted",
                                       "MultipleChunkSingleFloatColumn", "RepeatedInt", "RunNullValues", "SingleChunkSingleFloatColumn"])
def test_dictionary_encoder_reproduces_hyrise_bytes(directory, tmp_path):
    """<dir>/Unencoded.bin and <dir>/Dictionary.bin are the same table (BinaryWriterMultiEncodingTest writes both from one
    table, binary_writer_test.cpp:240-520).  Decode the unencoded file, dictionary-encode every numeric column HERE, and
    compare with what Hyrise's DictionaryEncoder produced: dictionary, attribute-vector width, value ids, NULL id."""
    plain = binary.read_table(os.path.join(BIN, directory, "Unencoded.bin"))
    encoded = binary.read_table(os.path.join(BIN, directory, "Dictionary.bin"))
    assert plain.names == encoded.names and plain.types == encoded.types
    for c in numeric_columns(plain):
        values, nulls = binary.decode_column(plain, c)
        their_values, their_nulls = binary.decode_column(encoded, c)
        np.testing.assert_array_equal(values, their_values)
        np.testing.assert_array_equal(nulls, their_nulls)
        for chunk, theirs in enumerate(encoded.columns[c].segments):
            if theirs.encoding != abi.ENC_DICTIONARY:   # a still-mutable last chunk stays unencoded (ChunkEncoder)
                continue
            begin = sum(s.size for s in encoded.columns[c].segments[:chunk])
            mask = nulls[begin:begin + theirs.size]
            ours = storage.encode_segment(values[begin:begin + theirs.size], mask if plain.nullable[c] else None, abi.ENC_DICTIONARY)
            assert ours.width == theirs.width and ours.aux_size == theirs.aux_size, (directory, c, chunk)
            np.testing.assert_array_equal(ours.aux, theirs.aux)
            np.testing.assert_array_equal(ours.data, theirs.data)


FOR_TABLES = [   # binary_writer_test.cpp:113-238: (file, chunk size, nullable, rows)
    ("SingleChunkFrameOfReferenceSegment.bin", 10, False, [1, 2, 3, 4, 5]),
    ("MultipleChunksFrameOfReferenceSegment.bin", 3, False, [1, 1, 2, 4, 5]),
    ("NullValuesFrameOfReferenceSegment.bin", 3, True, [1, None, 2, None, 5]),
    ("AllNullFrameOfReferenceSegment.bin", 3, True, [None] * 5),
]


@pytest.mark.parametrize("name,chunk_size,nullable,rows", FOR_TABLES, ids=[t[0] for t in FOR_TABLES])
def test_frame_of_reference_encoder_reproduces_hyrise_bytes(name, chunk_size, nullable, rows, tmp_path):
    values = np.array([0 if v is None else v for v in rows], dtype=np.int32)
    nulls = np.array([v is None for v in rows], dtype=bool) if nullable else None
    column = storage.make_column(values, nulls, abi.ENC_FRAME_OF_REFERENCE, chunk_size=chunk_size, nullable=nullable)
    table = binary.table_from_columns(["a"], [nullable], chunk_size, [column], [nulls])
    assert binary.write_table(str(tmp_path / "out.bin"), table) == open(os.path.join(BIN, name), "rb").read()
    parsed = binary.read_table(os.path.join(BIN, name))
    got_values, got_nulls = binary.decode_column(parsed, 0)
    np.testing.assert_array_equal(got_values, values)
    np.testing.assert_array_equal(got_nulls, nulls if nulls is not None else np.zeros(len(rows), dtype=bool))


def test_fixed_string_dictionary():
    """binary_writer_test.cpp FixedStringDictionarySingleChunk: "This", "is", "a", "test" in a FixedStringDictionarySegment."""
    table = binary.read_table(os.path.join(BIN, "FixedStringDictionarySingleChunk.bin"))
    segment, dictionary = table.columns[0].segments[0], table.strings[0][0]
    assert list(dictionary) == ["This", "a", "is", "test"] and dictionary.length == 4
    assert [dictionary[i] for i in segment.data] == ["This", "is", "a", "test"]
    assert segment.encoding == abi.ENC_DICTIONARY and segment.aux is None and segment.aux_size == 4


def test_known_table_contents():
    """int_float.bin is resources/test_data/tbl/int_float.tbl (binary_parser_test.cpp); TwoColumnsNoValues has no chunks."""
    table = binary.read_table(os.path.join(BIN, "int_float.bin"))
    assert table.names == ["a", "b"] and table.types == [abi.TYPE_INT, abi.TYPE_FLOAT] and table.chunk_size == 65535
    ints, _ = binary.decode_column(table, 0)
    floats, _ = binary.decode_column(table, 1)
    assert ints.tolist() == [12345, 123, 1234]
    np.testing.assert_allclose(floats, [458.7, 456.7, 457.7], rtol=1e-6)
    empty = binary.read_table(os.path.join(BIN, "TwoColumnsNoValues.bin"))
    assert empty.names == ["FirstColumn", "SecondColumn"] and empty.chunk_count == 0


@pytest.mark.parametrize("directory", ["AllTypesAllNullValues", "AllTypesMixColumn", "AllTypesNullValues", "AllTypesSegmentSorted", "AllTypesSegmentUnsorted",
                                       "RepeatedInt", "RunNullValues", "SingleChunkSingleFloatColumn"])
def test_run_length_encoder_reproduces_hyrise_bytes(directory):
    """Same game for RunLengthSegment: runs, NULL runs and inclusive end positions as Hyrise's RunLengthEncoder wrote them."""
    plain = binary.read_table(os.path.join(BIN, directory, "Unencoded.bin"))
    encoded = binary.read_table(os.path.join(BIN, directory, "RunLength.bin"))
    for c in numeric_columns(plain):
        values, nulls = binary.decode_column(plain, c)
        their_values, their_nulls = binary.decode_column(encoded, c)
        np.testing.assert_array_equal(values, their_values)
        np.testing.assert_array_equal(nulls, their_nulls)
        begin = 0
        for theirs in encoded.columns[c].segments:
            if theirs.encoding == abi.ENC_RUN_LENGTH:
                ours = storage.encode_run_length(values[begin:begin + theirs.size], nulls[begin:begin + theirs.size])
                assert ours.aux_size == theirs.aux_size, (directory, c)
                np.testing.assert_array_equal(ours.aux, theirs.aux)
                np.testing.assert_array_equal(ours.nulls, theirs.nulls)
                np.testing.assert_array_equal(ours.data[ours.nulls == 0], theirs.data[theirs.nulls == 0])   # a NULL run's value is unspecified
            begin += theirs.size
