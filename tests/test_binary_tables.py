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
EXPANDED = ("LZ4.bin", "LZ4MultipleBlocks.bin")   # read as the ValueSegments they were compressed from: no byte round trip
LZ4_DIRECTORIES = sorted(os.path.basename(os.path.dirname(p)) for p in FILES if p.endswith("/LZ4.bin"))


def test_fixtures_present():
    assert len(FILES) >= 40


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, BIN) for p in FILES])
def test_read_write_round_trip(path, tmp_path):
    if path.endswith(EXPANDED):
        pytest.skip("LZ4 segments are expanded while reading (test_lz4_* below)")
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


# ---- LZ4 segments and the BitPacking bit layout, pinned by files Hyrise wrote ---------------------------------------------------
@pytest.mark.parametrize("directory", LZ4_DIRECTORIES)
def test_lz4_segments_decode_to_their_unencoded_twins(directory, tmp_path):
    """<dir>/LZ4.bin and <dir>/Unencoded.bin are one table written twice (binary_writer_test.cpp:240-520).  Reading the LZ4 file
    -- LZ4 blocks, the zstd-trained dictionary as history, NULL vectors, and for strings the BitPacking offset vector -- gives the
    unencoded file's values, NULLs and strings; written back, the expanded table IS the unencoded file, byte for byte."""
    expanded, plain = binary.read_table(os.path.join(BIN, directory, "LZ4.bin")), binary.read_table(os.path.join(BIN, directory, "Unencoded.bin"))
    assert expanded.names == plain.names and expanded.types == plain.types and expanded.chunk_count == plain.chunk_count
    assert len(LZ4_DIRECTORIES) >= 11
    for c in range(len(plain.columns)):
        for chunk in range(plain.chunk_count):
            ours, theirs = expanded.null_masks[c][chunk], plain.null_masks[c][chunk]
            assert (ours is None or not ours.any()) == (theirs is None or not theirs.any())
            if theirs is not None and theirs.any():
                np.testing.assert_array_equal(ours, theirs)
            if plain.types[c] == abi.TYPE_STRING:
                valid = [i for i in range(len(plain.strings[c][chunk])) if theirs is None or not theirs[i]]
                assert [expanded.strings[c][chunk][i] for i in valid] == [plain.strings[c][chunk][i] for i in valid]
            else:
                keep = slice(None) if theirs is None else ~theirs
                a, b = np.asarray(expanded.columns[c].segments[chunk].data), np.asarray(plain.columns[c].segments[chunk].data)
                assert a.dtype == b.dtype and a[keep].tobytes() == b[keep].tobytes()
            assert expanded.columns[c].segments[chunk].encoding == abi.ENC_UNENCODED
    assert binary.write_table(str(tmp_path / "out.bin"), expanded) == open(os.path.join(BIN, directory, "Unencoded.bin"), "rb").read()


def test_lz4_multiple_blocks_known_answer():
    """binary_parser_test.cpp:247-268: 20 000 rows of four repeating tuples, every column LZ4-encoded in several 16 KiB blocks that
    share a dictionary; the strings come back through 18-bit BitPacking offsets."""
    table = binary.read_table(os.path.join(BIN, "LZ4MultipleBlocks.bin"))
    assert table.names == ["a", "b", "c", "d", "e"] and table.chunk_count == 1
    assert table.strings[0][0] == ["AAAAA", "BBBBBBBBBB", "CCCCCCCCCCCCCCC", "DDDDDDDDDDDDDDDDDDDD"] * 5000
    np.testing.assert_array_equal(table.columns[1].segments[0].data, np.tile(np.array([1, 2, 3, 4], dtype=np.int32), 5000))
    np.testing.assert_array_equal(table.columns[2].segments[0].data, np.tile(np.array([100, 200, 300, 400], dtype=np.int64), 5000))
    np.testing.assert_array_equal(table.columns[3].segments[0].data, np.tile(np.array([1.1, 2.2, 3.3, 4.4], dtype=np.float32), 5000))
    np.testing.assert_array_equal(table.columns[4].segments[0].data, np.tile(np.array([11.1, 22.2, 33.3, 44.4]), 5000))


def test_lz4_block_decoder_known_sequences():
    """lz4_Block_format.md by hand: literals only; a match that overlaps its own output (run length); 255-continued lengths; a match
    into the dictionary; corrupt offsets are refused."""
    assert binary.lz4_block_decode(bytes([0x50]) + b"hello", 5) == b"hello"
    assert binary.lz4_block_decode(bytes([0x1F, ord("a"), 1, 0, 10]) + bytes([0x00]), 30) == b"a" * 30           # 1 literal, match offset 1, length 4 + 15 + 10
    assert binary.lz4_block_decode(bytes([0xF0, 5]) + b"x" * 20, 20) == b"x" * 20                                  # literal length 15 + 5
    assert binary.lz4_block_decode(bytes([0x02, 6, 0]) + bytes([0x10]) + b"!", 7, history=b"abcdef") == b"abcdef!"   # 6 bytes from the dictionary
    with pytest.raises(ValueError):
        binary.lz4_block_decode(bytes([0x10, ord("a"), 9, 0]) + bytes([0x00]), 10)


def test_bit_packing_layout():
    """compact::vector<uint32_t, 0, uint64_t>: element i in bits [i b, (i + 1) b) of the little-endian word stream (the layout the
    LZ4 fixtures' string offsets pin); values that straddle words included."""
    rng = np.random.default_rng(8)
    for bits in (1, 3, 7, 8, 13, 18, 31, 32):
        values = rng.integers(0, 2 ** bits, 1000, dtype=np.uint64)
        stream = np.zeros((1000 * bits + 63) // 64 * 64, dtype=np.uint8)
        for i, v in enumerate(values):
            for b in range(bits):
                stream[i * bits + b] = (int(v) >> b) & 1
        words = np.packbits(stream, bitorder="little").view("<u8")
        np.testing.assert_array_equal(binary.unpack_bits(words, bits, 1000), values.astype(np.uint32))


def test_lz4_decoder_refuses_truncated_blocks_and_replicates_overlapping_matches():
    """lz4_block_decode checks every length against the block (a truncated block used to raise IndexError or silently copy a short
    literal run) and copies an overlapping match by repeating the offset-sized pattern (RLE-like blocks at interpreter speed before)."""
    # 4 literals "abcd", then a match of 20 bytes at offset 4 (overlapping: abcdabcd...), then the last literals "xy"
    block = bytes([0x4F]) + b"abcd" + bytes([4, 0]) + bytes([20 - 4 - 15]) + bytes([0x20]) + b"xy"
    want = b"abcd" + b"abcd" * 5 + b"xy"
    assert binary.lz4_block_decode(block, len(want)) == want
    run = bytes([0x1F]) + b"z" + bytes([1, 0]) + bytes([255, 255, 10]) + bytes([0x00])      # one literal, then 'z' 539 more times (offset 1)
    assert binary.lz4_block_decode(run, 1 + 4 + 15 + 255 + 255 + 10) == b"z" * 540
    for cut in (1, 3, 5, 6, 7):                                                              # every truncation is refused, never mis-decoded
        with pytest.raises(ValueError):
            binary.lz4_block_decode(block[:cut], len(want))
    with pytest.raises(ValueError):
        binary.lz4_block_decode(block, len(want) - 3)                                        # a match past the decoded size
    with pytest.raises(ValueError):
        binary.lz4_block_decode(bytes([0x00, 9, 0]), 4)                                      # offset outside the history
