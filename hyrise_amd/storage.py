"""Host-side column construction in Hyrise's segment layouts (numpy), and the handle of a device-resident column.

Restates, vectorised, what the reference's encoders produce so that synthetic and fixture data reaches the kernels in
exactly the layout Hyrise keeps in memory:
  DictionaryEncoder            src/lib/storage/dictionary_segment/dictionary_encoder.hpp:33-103
  FrameOfReferenceEncoder      src/lib/storage/frame_of_reference_segment/frame_of_reference_encoder.hpp:25-122
  FixedWidthIntegerCompressor  src/lib/storage/vector_compression/fixed_width_integer/fixed_width_integer_compressor.cpp:33-44
  auto_select_segment_encoding_spec  src/lib/storage/segment_encoding_utils.cpp:105-115
(tests/test_encoding.py checks these against the C oracle's scalar restatement of the same encoders.)
"""
import ctypes as C

import numpy as np

from . import abi

NP_TYPES = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}
TYPE_OF_NP = {np.dtype(v): k for k, v in NP_TYPES.items()}


def fixed_width(max_value):
    """Element width FixedWidthIntegerCompressor picks for a maximum value."""
    if max_value <= 0xFF:
        return 1
    if max_value <= 0xFFFF:
        return 2
    return 4


_UINT = {1: np.uint8, 2: np.uint16, 4: np.uint32}


def pack_nulls(nulls):
    """bool[n] -> libstdc++ vector<bool> storage: bit (i % 64) of u64 word (i / 64)."""
    n = len(nulls)
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = nulls
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


class HostSegment:
    """One segment with its buffers as numpy arrays (kept alive for as long as descriptors point at them)."""

    def __init__(self, encoding, data_type, size, width, data, aux=None, aux_size=0, nulls=None, ref=None,
                 ref_chunk_id=abi.INVALID_CHUNK_ID, sorted_by=0, bits=0):
        self.encoding, self.data_type, self.size, self.width = encoding, data_type, int(size), int(width)
        self.data, self.aux, self.aux_size, self.nulls = data, aux, int(aux_size), nulls
        self.ref, self.ref_chunk_id = ref, int(ref_chunk_id)
        self.sorted_by = int(sorted_by)   # abi.SORT_*: Chunk::individually_sorted_by names this column
        self.bits = int(bits)             # width == 0: `data` is a BitPackingVector of `bits` bits per element (uint64 words)
        self.lz4 = None                   # ENC_LZ4: (blocks as bytes objects, block_size, last_block_size, dictionary bytes) -- see lz4_descriptor

    def lz4_descriptor(self):
        """abi.Lz4Blocks over this segment's compressed blocks (and everything that must stay alive with it)."""
        import ctypes as C
        blocks, block_size, last_block_size, dictionary = self.lz4
        buffers = [np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8) for b in blocks]
        pointers = (C.c_void_p * max(1, len(blocks)))(*[b.ctypes.data for b in buffers])
        sizes = (C.c_uint32 * max(1, len(blocks)))(*[len(b) for b in blocks])
        dictionary_buffer = np.frombuffer(dictionary, dtype=np.uint8) if dictionary else None
        d = abi.Lz4Blocks()
        d.blocks, d.block_bytes = C.cast(pointers, C.POINTER(C.c_void_p)), C.cast(sizes, C.POINTER(C.c_uint32))
        d.block_count, d.block_size, d.last_block_size = len(blocks), block_size, last_block_size
        d.dictionary_bytes = len(dictionary) if dictionary else 0
        d.dictionary = dictionary_buffer.ctypes.data if dictionary_buffer is not None else None
        self._lz4_alive = (buffers, pointers, sizes, dictionary_buffer, d)
        return d


def encode_segment(values, nulls, encoding, data_type=None):
    """Encode one chunk's worth of values.  `nulls` is a bool array or None."""
    values = np.ascontiguousarray(values)
    data_type = data_type if data_type is not None else TYPE_OF_NP[values.dtype]
    n = len(values)
    has_nulls = nulls is not None and bool(np.any(nulls))
    if encoding == abi.ENC_UNENCODED:
        data = values.copy()
        if nulls is not None:  # a nullable ValueSegment keeps its null vector even without NULLs
            data[np.asarray(nulls, dtype=bool)] = 0  # NULL slots hold T{} (value_segment.hpp)
        words = pack_nulls(np.asarray(nulls, dtype=bool)) if nulls is not None else None
        return HostSegment(encoding, data_type, n, values.dtype.itemsize, data, nulls=words)
    if encoding == abi.ENC_DICTIONARY:
        mask = np.asarray(nulls, dtype=bool) if nulls is not None else np.zeros(n, dtype=bool)
        dense = values[~mask]
        if dense.dtype.kind == "f":
            # std::sort + std::unique use operator<,==: -0.0 == 0.0 collapse, np.unique agrees; NaNs are not supported
            assert not np.isnan(dense).any(), "NaN in a dictionary column"
        dictionary = np.unique(dense)
        if dictionary.dtype.kind == "f" and len(dictionary) > 1:
            keep = np.ones(len(dictionary), dtype=bool)
            keep[1:] = dictionary[1:] != dictionary[:-1]
            dictionary = dictionary[keep]
        d = len(dictionary)
        width = fixed_width(d)  # max value id == NULL value id == d (dictionary_encoder.hpp:85-92)
        av = np.full(n, d, dtype=np.uint32)
        av[~mask] = np.searchsorted(dictionary, dense, side="left")
        return HostSegment(encoding, data_type, n, width, av.astype(_UINT[width]), aux=dictionary.astype(values.dtype),
                           aux_size=d)
    if encoding == abi.ENC_FRAME_OF_REFERENCE:
        assert values.dtype == np.int32, "FrameOfReference is instantiated for int32 only"
        mask = np.asarray(nulls, dtype=bool) if nulls is not None else np.zeros(n, dtype=bool)
        blocks = (n + abi.FOR_BLOCK_SIZE - 1) // abi.FOR_BLOCK_SIZE
        padded = np.full(blocks * abi.FOR_BLOCK_SIZE, np.iinfo(np.int32).max, dtype=np.int64)
        padded[:n] = np.where(mask, np.iinfo(np.int32).max, values)
        minima = padded.reshape(blocks, abi.FOR_BLOCK_SIZE).min(axis=1).astype(np.int32) if blocks else np.zeros(0, np.int32)
        per_row_min = np.repeat(minima.astype(np.int64), abi.FOR_BLOCK_SIZE)[:n]
        offsets = np.where(mask, 0, values.astype(np.int64) - per_row_min).astype(np.uint32)
        width = fixed_width(int(offsets.max()) if n else 0)
        words = pack_nulls(mask) if has_nulls else None  # null vector only if a NULL was seen (:113-120)
        return HostSegment(encoding, data_type, n, width, offsets.astype(_UINT[width]), aux=minima, aux_size=blocks,
                           nulls=words)
    raise ValueError(f"encoding {encoding}")


def encode_string_dictionary(values, nulls=None):
    """DictionarySegment<pmr_string> of one chunk: the attribute vector goes to the device, the (byte-wise sorted,
    distinct) dictionary stays with the host, which resolves literals and LIKE patterns against it per chunk
    (dictionary_encoder.hpp:44-98).  Returns (HostSegment, dictionary as a list of bytes)."""
    raw = [v if isinstance(v, bytes) else str(v).encode("utf-8") for v in values]
    n = len(raw)
    mask = np.asarray(nulls, dtype=bool) if nulls is not None else np.zeros(n, dtype=bool)
    dictionary = sorted({v for v, is_null in zip(raw, mask) if not is_null})
    value_id = {v: i for i, v in enumerate(dictionary)}
    d = len(dictionary)
    width = fixed_width(d)
    av = np.array([d if is_null else value_id[v] for v, is_null in zip(raw, mask)], dtype=_UINT[width])
    return HostSegment(abi.ENC_DICTIONARY, abi.TYPE_STRING, n, width, av, aux=None, aux_size=d), dictionary


class HostColumn:
    """A column of a table, chunk by chunk."""

    def __init__(self, segments, data_type):
        self.segments = list(segments)
        self.data_type = data_type
        self._descriptors = None

    @property
    def n_chunks(self):
        return len(self.segments)

    @property
    def rows(self):
        return sum(s.size for s in self.segments)

    def descriptors(self, resolve_ref):
        """ctypes array of hy_segment with HOST pointers.  resolve_ref(HostColumn) -> pointer for `ref`."""
        arr = (abi.Segment * max(1, len(self.segments)))()
        for i, s in enumerate(self.segments):
            d = arr[i]
            d.encoding, d.data_type, d.size, d.width = s.encoding, s.data_type, s.size, s.width
            if s.encoding == abi.ENC_LZ4:
                import ctypes as C
                d.data = C.addressof(s.lz4_descriptor())
            else:
                d.data = s.data.ctypes.data if s.data is not None else None
            d.aux = s.aux.ctypes.data if s.aux is not None else None
            d.aux_size = s.aux_size
            d.ref_chunk_id = s.ref_chunk_id
            d.nulls = s.nulls.ctypes.data if s.nulls is not None else None
            d.ref = resolve_ref(s.ref) if s.ref is not None else None
            d.sorted_by, d.bits = s.sorted_by, s.bits
        return arr


def make_column(values, nulls=None, encoding=abi.ENC_UNENCODED, chunk_size=abi.CHUNK_DEFAULT_SIZE, nullable=None):
    """Split `values` into chunks of `chunk_size` rows (Table::append semantics) and encode each chunk.
    nullable: column definition's nullable flag; defaults to `nulls is not None`."""
    values = np.ascontiguousarray(values)
    n = len(values)
    nullable = (nulls is not None) if nullable is None else nullable
    segments = []
    for begin in range(0, n, chunk_size):
        end = min(n, begin + chunk_size)
        chunk_nulls = None
        if nulls is not None:
            chunk_nulls = np.asarray(nulls[begin:end], dtype=bool)
        elif nullable and encoding == abi.ENC_UNENCODED:
            chunk_nulls = np.zeros(end - begin, dtype=bool)
        segments.append(encode_segment(values[begin:end], chunk_nulls, encoding))
    return HostColumn(segments, TYPE_OF_NP[values.dtype])


def auto_encoding(data_type, unique):
    """auto_select_segment_encoding_spec (segment_encoding_utils.cpp:105-115)."""
    if unique:
        return abi.ENC_UNENCODED
    if data_type == abi.TYPE_INT:
        return abi.ENC_FRAME_OF_REFERENCE
    return abi.ENC_DICTIONARY


def encode_run_length(values, nulls=None):
    """RunLengthSegment<T> of one chunk (run_length_encoder.hpp): one run per stretch of equal values / NULLs; end positions
    are inclusive.  The NULL flags travel as one byte per run."""
    values = np.ascontiguousarray(values)
    n = len(values)
    mask = np.asarray(nulls, dtype=bool) if nulls is not None else np.zeros(n, dtype=bool)
    if n == 0:
        return HostSegment(abi.ENC_RUN_LENGTH, TYPE_OF_NP[values.dtype], 0, values.dtype.itemsize, values[:0].copy(), aux=np.zeros(0, np.uint32),
                           aux_size=0, nulls=np.zeros(0, np.uint8))
    clean = values.copy()
    clean[mask] = 0
    change = np.ones(n, dtype=bool)
    change[1:] = (clean[1:] != clean[:-1]) | (mask[1:] != mask[:-1])
    starts = np.flatnonzero(change)
    ends = np.concatenate([starts[1:] - 1, [n - 1]]).astype(np.uint32)
    return HostSegment(abi.ENC_RUN_LENGTH, TYPE_OF_NP[values.dtype], n, values.dtype.itemsize, clean[starts].copy(), aux=ends, aux_size=len(starts),
                       nulls=mask[starts].astype(np.uint8))


def pack_bits(values, bits):
    """BitPackingVector (compact::vector<uint32_t, 0, uint64_t>, bitpacking_vector_type.hpp:18): element i in bits
    [i * bits, (i + 1) * bits) of a little-endian stream of 64-bit words (the layout tests/test_binary_tables.py pins on files
    Hyrise wrote)."""
    values = np.asarray(values, dtype=np.uint64)
    n = len(values)
    words = (n * bits + 63) // 64
    stream = np.zeros(max(1, words) * 64, dtype=np.uint8)
    if n:
        stream[:n * bits] = ((values[:, None] >> np.arange(bits, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8).reshape(-1)
    return np.packbits(stream, bitorder="little").view(np.uint64)[:max(1, words)].copy()


def bit_pack_segment(segment):
    """The same Dictionary / FrameOfReference segment with its FixedWidthInteger vector as a BitPackingVector of the fewest bits
    that hold its largest element (bitpacking_compressor.cpp:14-30: at least one bit)."""
    assert segment.encoding in (abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE) and segment.width != 0
    top = int(segment.data.max()) if segment.size else 0
    bits = max(1, top.bit_length())
    return HostSegment(segment.encoding, segment.data_type, segment.size, 0, pack_bits(segment.data, bits), aux=segment.aux, aux_size=segment.aux_size,
                       nulls=segment.nulls, sorted_by=segment.sorted_by, bits=bits)


def expand_compressed(host_column):
    """The same column with every RunLength segment replaced by the ValueSegment it decodes to and every BitPackingVector by the
    FixedWidthInteger vector of the same elements (what the device decodes for the operators that gather rows; the CPU oracle
    reads plain segments only)."""
    if not any(s.encoding == abi.ENC_RUN_LENGTH or (s.width == 0 and s.bits) for s in host_column.segments):
        return host_column
    segments = []
    for s in host_column.segments:
        if s.width == 0 and s.bits and s.encoding != abi.ENC_RUN_LENGTH:
            from .binary import unpack_bits
            elements = unpack_bits(s.data, s.bits, s.size)
            width = 1 if s.bits <= 8 else 2 if s.bits <= 16 else 4
            segments.append(HostSegment(s.encoding, s.data_type, s.size, width, elements.astype(_UINT[width]), aux=s.aux, aux_size=s.aux_size, nulls=s.nulls,
                                        sorted_by=s.sorted_by))
            continue
        if s.encoding != abi.ENC_RUN_LENGTH:
            segments.append(s)
            continue
        lengths = np.diff(np.concatenate([[-1], np.asarray(s.aux, dtype=np.int64)]))
        mask = np.repeat(np.asarray(s.nulls, dtype=bool), lengths)
        values = np.repeat(s.data, lengths)
        values[mask] = 0
        segments.append(HostSegment(abi.ENC_UNENCODED, s.data_type, s.size, s.width, values, nulls=pack_nulls(mask) if mask.any() else None, sorted_by=s.sorted_by))
    return HostColumn(segments, host_column.data_type)


expand_run_length = expand_compressed   # (the name the tests of round 2 use)


MVCC_MUTABLE = 1 << 31
MAX_COMMIT_ID = 0xFFFFFFFF - 1   # MvccData::MAX_COMMIT_ID (mvcc_data.hpp): "not yet invalidated"


def make_mvcc_column(tids, begin_cids, end_cids, chunk_size=abi.CHUNK_DEFAULT_SIZE, mutable_chunks=(), invalid_row_counts=None):
    """A table's MvccData as a column of HY_ENC_MVCC segments (see hy_validate in include/hyrise_amd.h): per chunk the
    transaction ids, begin and end commit ids, max_begin_cid and the invalid-row counter (default: rows whose end_cid
    is set), bit 31 of which marks the chunk as still mutable."""
    tids, begin_cids, end_cids = (np.ascontiguousarray(a, dtype=np.uint32) for a in (tids, begin_cids, end_cids))
    n = len(tids)
    segments = []
    for chunk_id, begin in enumerate(range(0, n, chunk_size)):
        end = min(n, begin + chunk_size)
        b, e = begin_cids[begin:end].copy(), end_cids[begin:end].copy()
        committed = b[b < MAX_COMMIT_ID]
        max_begin = int(committed.max()) if len(committed) else 0
        if np.any(b >= MAX_COMMIT_ID):   # an uncommitted insert keeps max_begin_cid at its initial MAX_COMMIT_ID
            max_begin = MAX_COMMIT_ID
        invalid = int(np.sum(e < MAX_COMMIT_ID)) if invalid_row_counts is None else int(invalid_row_counts[chunk_id])
        flags = invalid | (MVCC_MUTABLE if chunk_id in mutable_chunks else 0)
        segments.append(HostSegment(abi.ENC_MVCC, abi.TYPE_INT, end - begin, 4, tids[begin:end].copy(), aux=b, aux_size=max_begin, nulls=e,
                                    ref_chunk_id=flags))
    return HostColumn(segments, abi.TYPE_INT)


def make_reference_column(referenced, pos_lists, single_chunk_ids=None):
    """Reference segments over `referenced` (a HostColumn of data segments).
    pos_lists: per chunk either an (n,2) uint32 array of (chunk_id, chunk_offset) RowIDs, or an int k meaning
    EntireChunkPosList of referenced chunk k.  single_chunk_ids[c]: the pos list's references_single_chunk() guarantee
    (common chunk id) or None."""
    segments = []
    for c, pos in enumerate(pos_lists):
        if isinstance(pos, (int, np.integer)):
            size = referenced.segments[int(pos)].size
            segments.append(HostSegment(abi.ENC_REFERENCE, referenced.data_type, size, 8, None, ref=referenced,
                                        ref_chunk_id=int(pos)))
        else:
            pos = np.ascontiguousarray(pos, dtype=np.uint32).reshape(-1, 2)
            common = single_chunk_ids[c] if single_chunk_ids is not None else None
            segments.append(HostSegment(abi.ENC_REFERENCE, referenced.data_type, len(pos), 8, pos, ref=referenced,
                                        ref_chunk_id=abi.INVALID_CHUNK_ID if common is None else int(common)))
    return HostColumn(segments, referenced.data_type)


class DeviceColumn:
    """RAII handle of an hy_column (the residency cache entry of one column)."""

    def __init__(self, host_column, refs=None):
        self.lib = abi.load_library()
        self.host = host_column
        self._refs = refs or {}

        def resolve(ref_host):
            return self._refs[id(ref_host)].handle

        self._descriptors = host_column.descriptors(resolve)
        handle = C.c_void_p()
        abi.check(self.lib.hy_column_create(self._descriptors, host_column.n_chunks, abi.MEM_HOST, C.byref(handle)))
        self.handle = handle
        self.n_chunks = host_column.n_chunks
        self.rows = host_column.rows
        self.data_type = host_column.data_type

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hy_column_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
