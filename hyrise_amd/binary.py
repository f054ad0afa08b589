"""Hyrise binary tables (`.bin`): reader and writer for the layouts this library consumes.

Format: src/lib/import_export/binary/binary_writer.hpp:24-215 (header, chunk header, one record per segment), written by
BinaryWriter::write (binary_writer.cpp) and read by BinaryParser (binary_parser.cpp).  A table exported by a Hyrise that
runs elsewhere -- exact dictionaries, attribute-vector widths, FrameOfReference blocks -- becomes `HostColumn`s whose
buffers go to the device unchanged (`storage.DeviceColumn`), and columns encoded here can be written back byte-identically.

Handled: Unencoded (ValueSegment), Dictionary and FrameOfReference segments with FixedWidthInteger attribute / offset
vectors and numeric RunLength segments, all five data types (string columns: parsed; only their dictionary-encoded form is
scannable on the device; FixedStringDictionary segments are string dictionaries for it).

LZ4 segments and BitPacking vectors are EXPANDED while reading, the way the residency cache expands RunLength segments: an
LZ4Segment becomes the ValueSegment it was compressed from (the reference itself can only decompress such a segment block by
block to scan it, lz4_segment.cpp:125-136), a BitPacking attribute / offset vector the FixedWidthInteger vector of the smallest
width that holds its values.  The LZ4 block format is the published one (lz4.org, lz4_Block_format.md; the reference links
liblz4, third_party/lz4: an absent submodule); the bit layout of the BitPacking container (compact::vector<uint32_t, 0, uint64_t>,
bitpacking_vector_type.hpp:18; github.com/gmarcais/compact_vector, also absent) is a little-endian bit stream -- element i in
bits [i b, (i + 1) b) of the 64-bit words -- PINNED here by files Hyrise wrote: the string offsets of every LZ4 string segment
under tests/golden/bin (12 files) decode to the strings of their Unencoded twins, and LZ4MultipleBlocks.bin to the table
binary_parser_test.cpp:247-268 spells out.
"""
import struct

import numpy as np

from . import abi
from .storage import HostColumn, HostSegment, pack_nulls

ENCODING_UNENCODED, ENCODING_DICTIONARY, ENCODING_RUN_LENGTH, ENCODING_FIXED_STRING, ENCODING_FRAME_OF_REFERENCE, ENCODING_LZ4 = range(6)  # encoding_type.hpp:26
VECTOR_BIT_PACKING, VECTOR_FIXED_1, VECTOR_FIXED_2, VECTOR_FIXED_4 = range(4)   # compressed_vector_type.hpp:28-33
TYPE_NAMES = {"int": abi.TYPE_INT, "long": abi.TYPE_LONG, "float": abi.TYPE_FLOAT, "double": abi.TYPE_DOUBLE, "string": abi.TYPE_STRING}
NAME_OF_TYPE = {v: k for k, v in TYPE_NAMES.items()}
NUMPY_OF_TYPE = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}
WIDTH_OF_VECTOR = {VECTOR_FIXED_1: 1, VECTOR_FIXED_2: 2, VECTOR_FIXED_4: 4}
UINT_OF_WIDTH = {1: np.uint8, 2: np.uint16, 4: np.uint32}


class UnsupportedSegment(Exception):
    pass


def lz4_block_decode(block, size, history=b""):
    """One LZ4 block (lz4_Block_format.md: token = literal length << 4 | match length - 4, 255-continued lengths, literals, 2-byte
    little-endian match offset; the last sequence ends after its literals) -> `size` bytes.  `history`: the dictionary the block was
    compressed with (LZ4_decompress_safe_usingDict, lz4_segment.cpp:217-220): matches may reach back into it."""
    out = bytearray(history)
    start, end, i, n = len(history), len(history) + size, 0, len(block)

    def continued(value, at):   # a length field of 15 continues in bytes of 255 ... < 255
        while True:
            if at >= n:
                raise ValueError("corrupt LZ4 block: a length runs past the end of the block")
            extra = block[at]
            at += 1
            value += extra
            if extra != 255:
                return value, at

    while i < n:
        token = block[i]
        i += 1
        literals = token >> 4
        if literals == 15:
            literals, i = continued(literals, i)
        if i + literals > n:
            raise ValueError("corrupt LZ4 block: literals run past the end of the block")   # (a slice past the end would silently be shorter)
        out += block[i:i + literals]
        i += literals
        if i >= n:
            break                                   # the last sequence has no match
        if i + 2 > n:
            raise ValueError("corrupt LZ4 block: truncated match offset")
        offset = block[i] | block[i + 1] << 8
        i += 2
        if offset == 0 or offset > len(out):
            raise ValueError("corrupt LZ4 block: match offset outside the history")
        length = (token & 15) + 4
        if (token & 15) == 15:
            length, i = continued(length, i)
        if len(out) + length > end:
            raise ValueError("corrupt LZ4 block: a match runs past the decoded size")
        position = len(out) - offset
        if offset >= length:
            out += out[position:position + length]
        else:                                       # overlapping match: the copied bytes feed the copy -- the last `offset` bytes repeat
            pattern = bytes(out[position:])
            out += (pattern * (length // offset + 1))[:length]
    if len(out) != end:
        raise ValueError(f"LZ4 block decoded to {len(out) - start} bytes, {size} expected")
    return bytes(out[start:])


def unpack_bits(words, bits, count):
    """compact::vector<uint32_t, 0, uint64_t>: `count` elements of `bits` bits each, a little-endian bit stream over 64-bit words.
    Element i = bits [i * bits, (i + 1) * bits): the low part from word i * bits / 64, the rest (if it straddles) from the next word."""
    if bits == 0 or count == 0:
        return np.zeros(count, dtype=np.uint32)
    words = np.concatenate([np.ascontiguousarray(words, dtype="<u8").astype(np.uint64), np.zeros(1, dtype=np.uint64)])   # (a spare word for the last element's "next word")
    at = np.arange(count, dtype=np.uint64) * np.uint64(bits)
    index, shift = (at >> np.uint64(6)).astype(np.int64), at & np.uint64(63)
    low = words[index] >> shift
    spill = shift + np.uint64(bits) > np.uint64(64)
    high = np.where(spill, words[index + 1] << ((np.uint64(64) - shift) & np.uint64(63)), np.uint64(0))
    return ((low | high) & np.uint64((1 << bits) - 1)).astype(np.uint32)


def fixed_width_of(values):
    """(width, vector) of the FixedWidthInteger vector that holds `values` (fixed_width_integer_compressor.cpp:33-44)."""
    top = int(values.max()) if len(values) else 0
    width = 1 if top <= 0xFF else 2 if top <= 0xFFFF else 4
    return width, values.astype(UINT_OF_WIDTH[width])


class FixedStrings(list):
    """Dictionary of a FixedStringDictionarySegment: the strings plus the fixed slot length they are stored with."""

    def __init__(self, values, length):
        super().__init__(values)
        self.length = length


class BinaryTable:
    """names / types (HY_TYPE_*) / nullable per column, target chunk size, and one HostColumn per column.  String
    columns keep their values as Python lists next to the HostColumn (`strings[column][chunk]`: dictionary or values)."""

    def __init__(self, names, types, nullable, chunk_size, columns, strings, null_masks, sort_definitions=None):
        self.names, self.types, self.nullable, self.chunk_size = names, types, nullable, chunk_size
        self.columns, self.strings, self.null_masks = columns, strings, null_masks
        self.sort_definitions = sort_definitions or []   # per chunk: list of (column id, sort mode) -- Chunk::individually_sorted_by

    @property
    def chunk_count(self):
        return self.columns[0].n_chunks if self.columns else 0


class _Reader:
    def __init__(self, data):
        self.data, self.pos = data, 0

    def take(self, fmt):
        size = struct.calcsize(fmt)
        values = struct.unpack_from("<" + fmt, self.data, self.pos)
        self.pos += size
        return values if len(values) > 1 else values[0]

    def array(self, dtype, count):
        out = np.frombuffer(self.data, dtype=dtype, count=count, offset=self.pos).copy()
        self.pos += out.nbytes
        return out

    def compact_vector(self, count):   # export_compact_vector (binary_writer.cpp:106-109): bit width, then the raw 64-bit words
        bits = self.take("B")
        words = (count * bits + 63) // 64
        return unpack_bits(self.array(np.dtype("<u8"), words), bits, count)

    def vector(self, vector_type, count):
        """An attribute / offset vector of any CompressedVectorType -> (width, FixedWidthInteger values)."""
        if vector_type == VECTOR_BIT_PACKING:
            return fixed_width_of(self.compact_vector(count))
        width = WIDTH_OF_VECTOR[vector_type]
        return width, self.array(UINT_OF_WIDTH[width], count)

    def strings(self, count):   # export_string_values (binary_writer.cpp:48-76): size_t lengths, then the bytes back to back
        lengths = self.array(np.uint64, count)
        out = []
        for n in lengths:
            out.append(self.data[self.pos:self.pos + int(n)].decode("utf-8", errors="surrogateescape"))
            self.pos += int(n)
        return out


def read_table(path, keep_lz4=False):
    """keep_lz4: numeric LZ4 segments stay compressed (abi.ENC_LZ4: the library decompresses them on the device); their decoded twin is kept
    in `segment.decoded` for comparison."""
    with open(path, "rb") as fh:
        r = _Reader(fh.read())
    chunk_size, chunk_count, column_count = r.take("I"), r.take("I"), r.take("H")
    types = [TYPE_NAMES[t] for t in r.strings(column_count)]
    nullable = [bool(b) for b in r.array(np.uint8, column_count)]
    names = r.strings(column_count)
    segments = [[] for _ in range(column_count)]
    strings = [[] for _ in range(column_count)]
    null_masks = [[] for _ in range(column_count)]
    sort_definitions = []
    for _ in range(chunk_count):
        rows = r.take("I")
        sort_definitions.append([r.take("HB") for _ in range(r.take("I"))])   # SortColumnDefinition: ColumnID (2) + SortMode (1)
        for c in range(column_count):
            segment, text, nulls = _read_segment(r, types[c], nullable[c], rows, keep_lz4)
            segments[c].append(segment)
            strings[c].append(text)
            null_masks[c].append(nulls)
    if r.pos != len(r.data):
        raise ValueError(f"{path}: {len(r.data) - r.pos} trailing bytes")
    columns = [HostColumn(segments[c], types[c]) for c in range(column_count)]
    return BinaryTable(names, types, nullable, chunk_size, columns, strings, null_masks, sort_definitions)


def _read_segment(r, data_type, column_nullable, rows, keep_lz4=False):
    encoding = r.take("B")
    is_string = data_type == abi.TYPE_STRING
    if encoding == ENCODING_UNENCODED:                   # binary_writer.hpp:56-76
        nulls = None
        if column_nullable and r.take("B"):
            nulls = r.array(np.uint8, rows).astype(bool)
        if is_string:
            values = r.strings(rows)
            return HostSegment(abi.ENC_UNENCODED, data_type, rows, 0, None), values, nulls
        values = r.array(NUMPY_OF_TYPE[data_type], rows)
        words = pack_nulls(nulls) if nulls is not None else None
        return HostSegment(abi.ENC_UNENCODED, data_type, rows, values.dtype.itemsize, values, nulls=words), None, nulls
    if encoding == ENCODING_DICTIONARY:                  # binary_writer.hpp:98-120
        vector_type = r.take("B")
        dictionary_size = r.take("I")
        text = None
        if is_string:
            text = r.strings(dictionary_size)
            dictionary = None
        else:
            dictionary = r.array(NUMPY_OF_TYPE[data_type], dictionary_size)
        width, attribute_vector = r.vector(vector_type, rows)
        nulls = attribute_vector == dictionary_size     # NULL value id (dictionary_segment.cpp:139-141)
        return (HostSegment(abi.ENC_DICTIONARY, data_type, rows, width, attribute_vector, aux=dictionary, aux_size=dictionary_size), text,
                nulls if nulls.any() else None)
    if encoding == ENCODING_FIXED_STRING:                # binary_writer.hpp:122-144: a dictionary of fixed-length char slots
        vector_type = r.take("B")
        dictionary_size, length = r.take("I"), r.take("I")
        raw = r.data[r.pos:r.pos + dictionary_size * length]
        r.pos += dictionary_size * length
        text = FixedStrings([raw[i * length:(i + 1) * length].rstrip(b"\0").decode("utf-8", errors="surrogateescape") for i in range(dictionary_size)], length)
        width, attribute_vector = r.vector(vector_type, rows)
        nulls = attribute_vector == dictionary_size
        # for the device this IS a string dictionary segment: scans compare value ids the caller resolved
        return HostSegment(abi.ENC_DICTIONARY, data_type, rows, width, attribute_vector, aux=None, aux_size=dictionary_size), text, nulls if nulls.any() else None
    if encoding == ENCODING_FRAME_OF_REFERENCE:          # binary_writer.hpp:168-190
        vector_type = r.take("B")
        blocks = r.take("I")
        minima = r.array(np.int32, blocks)
        nulls = r.array(np.uint8, rows).astype(bool) if r.take("B") else None
        width, offsets = r.vector(vector_type, rows)
        words = pack_nulls(nulls) if nulls is not None else None
        return HostSegment(abi.ENC_FRAME_OF_REFERENCE, data_type, rows, width, offsets, aux=minima, aux_size=blocks, nulls=words), None, nulls
    if encoding == ENCODING_RUN_LENGTH:                  # binary_writer.hpp:146-160
        runs = r.take("I")
        text = r.strings(runs) if is_string else None
        run_values = None if is_string else r.array(NUMPY_OF_TYPE[data_type], runs)
        run_nulls = r.array(np.uint8, runs)
        run_ends = r.array(np.uint32, runs)
        lengths = np.diff(np.concatenate([[-1], run_ends.astype(np.int64)]))
        nulls = np.repeat(run_nulls.astype(bool), lengths)
        width = 0 if is_string else run_values.dtype.itemsize
        return (HostSegment(abi.ENC_RUN_LENGTH, data_type, rows, width, run_values, aux=run_ends, aux_size=runs, nulls=run_nulls), text,
                nulls if nulls.any() else None)
    if encoding == ENCODING_LZ4:                         # binary_writer.hpp:194-223, binary_parser.cpp:263-305
        elements, block_count, block_size, last_block_size = r.take("I"), r.take("I"), r.take("I"), r.take("I")
        block_sizes = r.array(np.uint32, block_count)
        blocks = []
        for size in block_sizes:
            blocks.append(bytes(r.data[r.pos:r.pos + int(size)]))
            r.pos += int(size)
        null_count = r.take("I")
        nulls = r.array(np.uint8, null_count).astype(bool) if null_count else None
        dictionary_bytes = r.take("I")
        dictionary = bytes(r.data[r.pos:r.pos + dictionary_bytes])
        r.pos += dictionary_bytes
        offset_count = r.take("I")
        offsets = r.compact_vector(rows) if offset_count else None
        # every block was compressed on its own (with the dictionary, if there is one): lz4_segment.cpp:191-226
        raw = b"".join(lz4_block_decode(block, block_size if b + 1 < block_count else last_block_size, dictionary) for b, block in enumerate(blocks))
        if nulls is not None and not nulls.any():
            nulls = None
        if is_string:
            if offsets is None:                          # only empty strings: nothing was compressed (lz4_segment.cpp:141-147)
                values = [""] * rows
            else:
                ends = np.concatenate([offsets[1:].astype(np.int64), [len(raw)]])
                values = [raw[int(b):int(e)].decode("utf-8", errors="surrogateescape") for b, e in zip(offsets, ends)]
            return HostSegment(abi.ENC_UNENCODED, data_type, rows, 0, None), values, nulls
        values = np.frombuffer(raw, dtype=NUMPY_OF_TYPE[data_type], count=elements).copy()
        words = pack_nulls(nulls) if nulls is not None else None
        if keep_lz4:   # the segment as Hyrise holds it: the library's device decoder gets the blocks
            segment = HostSegment(abi.ENC_LZ4, data_type, rows, values.dtype.itemsize, None, nulls=words)
            segment.lz4 = (blocks, block_size, last_block_size, dictionary)
            segment.decoded = values
            return segment, None, nulls
        return HostSegment(abi.ENC_UNENCODED, data_type, rows, values.dtype.itemsize, values, nulls=words), None, nulls
    raise UnsupportedSegment(f"encoding {encoding}")


def _write_strings(out, values):
    raw = [v.encode("utf-8", errors="surrogateescape") for v in values]
    out.append(np.array([len(v) for v in raw], dtype=np.uint64).tobytes())
    out.extend(raw)


def write_table(path, table):
    """Serialises a BinaryTable exactly like BinaryWriter::write: tables parsed from a file, or built from columns that
    this package encoded, come out byte-identical to what Hyrise writes."""
    out = [struct.pack("<IIH", table.chunk_size, table.chunk_count, len(table.names))]
    _write_strings(out, [NAME_OF_TYPE[t] for t in table.types])
    out.append(bytes(int(n) for n in table.nullable))
    _write_strings(out, table.names)
    for chunk in range(table.chunk_count):
        rows = table.columns[0].segments[chunk].size
        sorted_by = table.sort_definitions[chunk] if chunk < len(table.sort_definitions) else []
        out.append(struct.pack("<II", rows, len(sorted_by)))
        for column_id, mode in sorted_by:
            out.append(struct.pack("<HB", column_id, mode))
        for c, column in enumerate(table.columns):
            s = column.segments[chunk]
            text = table.strings[c][chunk] if table.strings[c] else None
            nulls = table.null_masks[c][chunk] if table.null_masks[c] else None
            if s.encoding == abi.ENC_UNENCODED:
                out.append(struct.pack("<B", ENCODING_UNENCODED))
                if table.nullable[c]:
                    # a nullable column's ValueSegment is nullable itself (value_segment.hpp), even without NULLs
                    mask = nulls if nulls is not None else np.zeros(rows, dtype=bool)
                    out.append(struct.pack("<B", 1))
                    out.append(mask.astype(np.uint8).tobytes())
                if table.types[c] == abi.TYPE_STRING:
                    _write_strings(out, text)
                else:
                    out.append(np.ascontiguousarray(s.data).tobytes())
            elif s.encoding == abi.ENC_DICTIONARY and isinstance(text, FixedStrings):
                out.append(struct.pack("<BBII", ENCODING_FIXED_STRING, {1: VECTOR_FIXED_1, 2: VECTOR_FIXED_2, 4: VECTOR_FIXED_4}[s.width], s.aux_size, text.length))
                out.extend(v.encode("utf-8", errors="surrogateescape").ljust(text.length, b"\0") for v in text)
                out.append(np.ascontiguousarray(s.data).tobytes())
            elif s.encoding == abi.ENC_DICTIONARY:
                out.append(struct.pack("<BBI", ENCODING_DICTIONARY, {1: VECTOR_FIXED_1, 2: VECTOR_FIXED_2, 4: VECTOR_FIXED_4}[s.width], s.aux_size))
                if table.types[c] == abi.TYPE_STRING:
                    _write_strings(out, text)
                else:
                    out.append(np.ascontiguousarray(s.aux).tobytes())
                out.append(np.ascontiguousarray(s.data).tobytes())
            elif s.encoding == abi.ENC_FRAME_OF_REFERENCE:
                out.append(struct.pack("<BBI", ENCODING_FRAME_OF_REFERENCE, {1: VECTOR_FIXED_1, 2: VECTOR_FIXED_2, 4: VECTOR_FIXED_4}[s.width], s.aux_size))
                out.append(np.ascontiguousarray(s.aux).tobytes())
                out.append(struct.pack("<B", 1 if nulls is not None else 0))
                if nulls is not None:
                    out.append(nulls.astype(np.uint8).tobytes())
                out.append(np.ascontiguousarray(s.data).tobytes())
            elif s.encoding == abi.ENC_RUN_LENGTH:
                out.append(struct.pack("<BI", ENCODING_RUN_LENGTH, s.aux_size))
                if table.types[c] == abi.TYPE_STRING:
                    _write_strings(out, text)
                else:
                    out.append(np.ascontiguousarray(s.data).tobytes())
                out.append(np.ascontiguousarray(s.nulls, dtype=np.uint8).tobytes())
                out.append(np.ascontiguousarray(s.aux, dtype=np.uint32).tobytes())
            else:
                raise UnsupportedSegment(f"encoding {s.encoding}")
    data = b"".join(out)
    with open(path, "wb") as fh:
        fh.write(data)
    return data


def decode_column(table, column):
    """(values, null mask) of a numeric column over all chunks, decoded on the host (for tests)."""
    values, masks = [], []
    for chunk, s in enumerate(table.columns[column].segments):
        nulls = table.null_masks[column][chunk]
        mask = nulls if nulls is not None else np.zeros(s.size, dtype=bool)
        if s.encoding == abi.ENC_UNENCODED:
            v = s.data.copy()
        elif s.encoding == abi.ENC_RUN_LENGTH:
            v = np.repeat(s.data, np.diff(np.concatenate([[-1], s.aux.astype(np.int64)])))
        elif s.encoding == abi.ENC_DICTIONARY:
            padded = np.concatenate([s.aux, np.zeros(1, dtype=s.aux.dtype)])
            v = padded[np.minimum(s.data.astype(np.int64), s.aux_size)]
        else:
            v = (s.data.astype(np.int64) + np.repeat(s.aux.astype(np.int64), abi.FOR_BLOCK_SIZE)[:s.size]).astype(np.int32)
        v = v.copy()
        v[mask] = 0
        values.append(v)
        masks.append(mask)
    if not values:
        return np.zeros(0, dtype=NUMPY_OF_TYPE[table.types[column]]), np.zeros(0, dtype=bool)
    return np.concatenate(values), np.concatenate(masks)


def table_from_columns(names, nullable, chunk_size, columns, nulls=None):
    """BinaryTable over HostColumns encoded by storage.make_column (numeric columns).  nulls[c]: the column's bool array
    or None.  The per-chunk NULL masks are what the writer needs beside the segments: a nullable column's ValueSegment
    always carries its null vector, a FrameOfReferenceSegment only if the chunk holds a NULL."""
    nulls = nulls or [None] * len(columns)
    masks = []
    for c, column in enumerate(columns):
        per_chunk, begin = [], 0
        for s in column.segments:
            mask = np.asarray(nulls[c][begin:begin + s.size], dtype=bool) if nulls[c] is not None else None
            begin += s.size
            if s.encoding == abi.ENC_UNENCODED:
                per_chunk.append(mask if mask is not None else (np.zeros(s.size, dtype=bool) if nullable[c] else None))
            elif s.encoding == abi.ENC_FRAME_OF_REFERENCE:
                per_chunk.append(mask if s.nulls is not None else None)
            else:
                per_chunk.append(mask if mask is not None and mask.any() else None)
        masks.append(per_chunk)
    return BinaryTable(list(names), [c.data_type for c in columns], list(nullable), chunk_size, list(columns), [[] for _ in columns], masks)
