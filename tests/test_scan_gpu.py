"""Parity of the HIP TableScan with the CPU oracle: PosLists, per-chunk offsets/counts and early-out states must be
bit-identical.  Every call goes through the C ABI (hy_table_scan / hy_table_scan_columns)."""
import ctypes as C

import numpy as np
import pytest

from golden import known_answers as KA
from hyrise_amd import abi, storage
from hyrise_amd.operators import make_predicate, table_scan, table_scan_columns
from hyrise_amd.storage import DeviceColumn
from support import (DeviceArray, assert_scan_equal, build_column, decode_rows, load_tbl, oracle_scan, oracle_scan_columns,
                     result_rows)

pytestmark = pytest.mark.gpu

ENCODINGS = [abi.ENC_UNENCODED, abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE]
CONDITIONS = [abi.PRED_EQUALS, abi.PRED_NOT_EQUALS, abi.PRED_LESS_THAN, abi.PRED_LESS_THAN_EQUALS,
              abi.PRED_GREATER_THAN, abi.PRED_GREATER_THAN_EQUALS, abi.PRED_BETWEEN_INCLUSIVE,
              abi.PRED_BETWEEN_LOWER_EXCLUSIVE, abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.PRED_BETWEEN_EXCLUSIVE,
              abi.PRED_IS_NULL, abi.PRED_IS_NOT_NULL]


def check(host_column, predicate, device_column=None, flags=0, excluded=None, context=""):
    dev = device_column or DeviceColumn(host_column)
    got = table_scan(dev, predicate, excluded_chunks=excluded, flags=flags)
    if excluded is None:
        want = oracle_scan(host_column, predicate, flags=flags)
        assert_scan_equal(got, want, context)
    return got


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_scan_matches_oracle_small(device, np_type, with_nulls):
    rng = np.random.default_rng(3)
    data_type = storage.TYPE_OF_NP[np.dtype(np_type)]
    for n, chunk in ((5000, 777), (20000, 8200), (1, 10), (8192, 8192), (8193, 65535)):
        values = rng.integers(-50, 50, n).astype(np_type)
        nulls = (rng.random(n) < 0.1) if with_nulls else None
        for encoding in ENCODINGS:
            if encoding == abi.ENC_FRAME_OF_REFERENCE and np_type != np.int32:
                continue
            host = build_column(values, nulls, chunk, encoding, nullable=with_nulls)
            dev = DeviceColumn(host)
            for condition in CONDITIONS:
                for value, value2 in ((-7, 12), (3, 3), (12, -7), (-1000, 1000), (49, 49), (-50, -50)):
                    for flags in (0, abi.SCAN_MATERIALIZE_ALL_MATCH):
                        p = make_predicate(condition, data_type, value, value2, nullable=with_nulls)
                        check(host, p, dev, flags, context=f"type {np_type.__name__} enc {encoding} cond {condition} "
                                                           f"lit {value},{value2} n {n} chunk {chunk} flags {flags}")


@pytest.mark.parametrize("width_values", [200, 3000, 70000])
def test_scan_attribute_vector_widths(device, width_values):
    """u8 / u16 / u32 attribute vectors and FoR offsets (fixed_width_integer_compressor.cpp:33-44)."""
    rng = np.random.default_rng(5)
    n = 300_000
    values = rng.integers(0, width_values, n).astype(np.int32)
    nulls = rng.random(n) < 0.02
    chunk = 65535 if width_values <= 65535 else 290_000   # > 65536 distinct values need a bigger chunk
    for encoding in (abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE):
        for with_nulls in (False, True):
            host = build_column(values, nulls if with_nulls else None, chunk, encoding, nullable=with_nulls)
            expected_width = 1 if width_values <= 255 else 2 if width_values <= 65535 else 4
            assert host.segments[0].width == expected_width
            dev = DeviceColumn(host)
            for condition in CONDITIONS:
                p = make_predicate(condition, abi.TYPE_INT, width_values // 3, 2 * width_values // 3, nullable=with_nulls)
                check(host, p, dev, context=f"enc {encoding} cond {condition} width {expected_width}")


def test_reference_fixtures_on_device(device):
    """The reference's inline known answers (table_scan_test.cpp:407-431,486-534) through the HIP path."""
    for encoding in ENCODINGS:
        for path, chunk in (("int_int_shuffled.tbl", 7), ("int_int_shuffled_2.tbl", 5)):
            t = load_tbl(path)
            a = build_column(t.columns[0], None, chunk, [encoding, encoding])
            b = build_column(t.columns[1], None, chunk, [encoding, encoding])
            dev = DeviceColumn(a)
            for answers, literal in ((KA.SCAN_ON_COMPRESSED_SEGMENTS, 6), (KA.SCAN_VALUE_GREATER_THAN_MAX, 30),
                                     (KA.SCAN_VALUE_LESS_THAN_MIN, -10)):
                for condition, expected in answers.items():
                    got = check(a, make_predicate(condition, abi.TYPE_INT, literal), dev)
                    assert sorted(decode_rows(b, result_rows(got))) == sorted(expected)


def test_scan_for_null_values_on_device(device):
    t = load_tbl("int_int_w_null_8_rows.tbl")
    for encoding in ENCODINGS:
        a = build_column(t.columns[0], t.nulls[0], 4, encoding)
        b = build_column(t.columns[1], t.nulls[1], 4, encoding)
        for condition, expected in KA.SCAN_FOR_NULL_VALUES.items():
            got = check(b, make_predicate(condition, abi.TYPE_INT, nullable=True))
            values = decode_rows(a, result_rows(got))
            assert sorted(values, key=lambda v: (v is None, v)) == sorted(expected, key=lambda v: (v is None, v))


def test_wide_dictionary_on_device(device):
    for entries, literal, expected_rows in KA.WIDE_DICTIONARY:
        host = build_column(np.arange(entries + 1, dtype=np.int32), None, 100_000, abi.ENC_DICTIONARY)
        got = check(host, make_predicate(abi.PRED_GREATER_THAN, abi.TYPE_INT, literal))
        assert got.total == expected_rows


def test_scan_reference_segments_single_chunk(device):
    """Reference tables as a first TableScan (or Validate's EntireChunkPosList, validate.cpp:275) produces them:
    every pos list references one chunk (fast path, abstract_dereferenced_column_table_scan_impl.cpp:38-46)."""
    rng = np.random.default_rng(9)
    n, chunk = 100_000, 20_000
    values = rng.integers(0, 1000, n).astype(np.int32)
    nulls = rng.random(n) < 0.05
    for encoding in ENCODINGS:
        base = build_column(values, nulls, chunk, encoding)
        first = oracle_scan(base, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 700, nullable=True),
                            flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        pos_lists, single = [], []
        for c in range(base.n_chunks):
            pos_lists.append(first.pos_list(c).copy())
            single.append(c)
        pos_lists.append(2)  # EntireChunkPosList over chunk 2
        single.append(2)
        ref_host = storage.make_reference_column(base, pos_lists, single)
        base_dev = DeviceColumn(base)
        ref_dev = DeviceColumn(ref_host, refs={id(base): base_dev})
        for condition in CONDITIONS:
            p = make_predicate(condition, abi.TYPE_INT, 100, 400, nullable=True)
            check(ref_host, p, ref_dev, context=f"reference enc {encoding} cond {condition}")


def test_multi_chunk_pos_lists(device):
    """Pos lists that reference several chunks (a scan on a join's output): the reference splits them by referenced
    chunk and appends the matches sub-list by sub-list (abstract_dereferenced_column_table_scan_impl.cpp:49-106,
    split_pos_list_by_chunk_id.cpp:13-59) -- matches come out ordered by (referenced chunk, position), NULL_ROW_IDs
    last.  The reference's own weird-pos-list test (operators/table_scan_test.cpp) plus random shuffled pos lists with
    NULL_ROW_IDs over every encoding."""
    t = load_tbl("int_int_shuffled_2.tbl")
    a = build_column(t.columns[0], None, 5, abi.ENC_DICTIONARY)
    ref = storage.make_reference_column(a, [np.array(KA.WEIRD_POS_LIST, dtype=np.uint32)], [None])
    base_dev = DeviceColumn(a)
    ref_dev = DeviceColumn(ref, refs={id(a): base_dev})
    for condition in CONDITIONS:
        check(ref, make_predicate(condition, abi.TYPE_INT, 10, 100, nullable=True), ref_dev, context=f"weird pos list cond {condition}")

    rng = np.random.default_rng(17)
    n, chunk = 200_000, 700   # 286 referenced chunks: two radix passes
    values = rng.integers(0, 1000, n).astype(np.int32)
    nulls = rng.random(n) < 0.05
    for encoding in ENCODINGS:
        base = build_column(values, nulls, chunk, encoding)
        base_dev = DeviceColumn(base)
        pos_lists = []
        for size in (0, 1, 63, 64, 65, 30_000, 65_535):
            rows = rng.integers(0, n, size)
            pos = np.stack([rows // chunk, rows % chunk], axis=1).astype(np.uint32)
            pos[rng.random(size) < 0.03] = 0xFFFFFFFF   # NULL_ROW_IDs (outer joins)
            pos_lists.append(pos)
        ref_host = storage.make_reference_column(base, pos_lists, [None] * len(pos_lists))
        ref_dev = DeviceColumn(ref_host, refs={id(base): base_dev})
        for condition in CONDITIONS:
            p = make_predicate(condition, abi.TYPE_INT, 100, 400, nullable=True)
            check(ref_host, p, ref_dev, context=f"multi-chunk pos list enc {encoding} cond {condition}")


def test_column_vs_column(device):
    rng = np.random.default_rng(13)
    n, chunk = 70_000, 9_000
    left = rng.integers(0, 20, n).astype(np.int32)
    rights = [rng.integers(0, 20, n).astype(np.int32), rng.integers(0, 20, n).astype(np.int64),
              (rng.integers(0, 20, n) + 0.5 * (rng.random(n) < 0.5)).astype(np.float32),
              (rng.integers(0, 20, n) + 0.5 * (rng.random(n) < 0.5)).astype(np.float64)]
    lnull, rnull = rng.random(n) < 0.1, rng.random(n) < 0.1
    for right in rights:
        for lenc in ENCODINGS:
            for renc in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
                lcol, rcol = build_column(left, lnull, chunk, lenc), build_column(right, rnull, chunk, renc)
                ldev, rdev = DeviceColumn(lcol), DeviceColumn(rcol)
                for condition in CONDITIONS[:6]:
                    got = table_scan_columns(ldev, rdev, condition)
                    want = oracle_scan_columns(lcol, rcol, condition)
                    assert_scan_equal(got, want, f"col-vs-col {right.dtype} lenc {lenc} renc {renc} cond {condition}")


def test_column_vs_column_four_byte_rows(device):
    """The wide path of ColumnVsColumn: int32 against int32 and float against float, every pair of encodings (FrameOfReference on
    either side of the ints; 1-, 2- and 4-byte value ids), NULLs on either side, no NULLs at all, negative values, NaNs, chunk
    sizes that are no multiple of eight, a NULL-only chunk."""
    rng = np.random.default_rng(29)
    n, chunk = 150_001, 20_003
    nulls = [(None, None), (rng.random(n) < 0.1, None), (rng.random(n) < 0.05, rng.random(n) < 0.2)]
    for spread in (10, 300, 100_000):      # dictionary sizes: u8, u16 and u32 value ids
        left, right = rng.integers(-spread, spread, n).astype(np.int32), rng.integers(-spread, spread, n).astype(np.int32)
        for lnull, rnull in nulls:
            if lnull is not None:
                lnull = lnull.copy()
                lnull[chunk:2 * chunk] = True                                   # chunk 1 of the left column: only NULLs
            for lenc in ENCODINGS:
                for renc in ENCODINGS:
                    lcol, rcol = build_column(left, lnull, chunk, lenc), build_column(right, rnull, chunk, renc)
                    ldev, rdev = DeviceColumn(lcol), DeviceColumn(rcol)
                    for condition in CONDITIONS[:6]:
                        assert_scan_equal(table_scan_columns(ldev, rdev, condition), oracle_scan_columns(lcol, rcol, condition),
                                          f"int32 spread {spread} lenc {lenc} renc {renc} cond {condition}")
    lf = (rng.integers(-50, 50, n) * 0.25).astype(np.float32)
    rf = (rng.integers(-50, 50, n) * 0.25).astype(np.float32)
    rf[rng.random(n) < 0.01] = -0.0
    with_nan = lf.copy()
    with_nan[rng.random(n) < 0.01] = np.nan                                      # (dictionaries cannot hold NaNs: unencoded only)
    lcol, rcol = build_column(with_nan, None, chunk, abi.ENC_UNENCODED), build_column(rf, nulls[1][0], chunk, abi.ENC_UNENCODED)
    for condition in CONDITIONS[:6]:
        assert_scan_equal(table_scan_columns(DeviceColumn(lcol), DeviceColumn(rcol), condition), oracle_scan_columns(lcol, rcol, condition), f"float NaN cond {condition}")
    for lnull, rnull in nulls:
        for lenc in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
            for renc in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
                lcol, rcol = build_column(lf, lnull, chunk, lenc), build_column(rf, rnull, chunk, renc)
                ldev, rdev = DeviceColumn(lcol), DeviceColumn(rcol)
                for condition in CONDITIONS[:6]:
                    assert_scan_equal(table_scan_columns(ldev, rdev, condition), oracle_scan_columns(lcol, rcol, condition),
                                      f"float lenc {lenc} renc {renc} cond {condition}")


@pytest.mark.parametrize("two_columns", [1, 0], ids=["two_stream_kernel", "generic_instantiation"])
def test_column_vs_column_dates(device, options, two_columns):
    """TPC-H Q4 / Q12's l_commitdate < l_receiptdate: two dictionary columns with u16 value ids (the chunks' dictionaries staged in LDS by
    scan_two_columns), a FrameOfReference twin, full and ragged chunks -- and the same scans with the kernel switched off (the generic
    instantiation every other shape takes): both against the oracle."""
    options.set(abi.OPT_SCAN_TWO_COLUMNS, two_columns)
    rng = np.random.default_rng(412)
    n = 65_535 * 3 + 4_321
    orderdate = rng.integers(0, 2406, n, dtype=np.int32)
    commit = (orderdate + rng.integers(30, 91, n, dtype=np.int32)).astype(np.int32)
    receipt = (orderdate + rng.integers(2, 152, n, dtype=np.int32)).astype(np.int32)
    receipt_nulls = rng.random(n) < 0.02
    for chunk in (65_535, 20_000):
        for lenc, renc in ((abi.ENC_DICTIONARY, abi.ENC_DICTIONARY), (abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE), (abi.ENC_FRAME_OF_REFERENCE, abi.ENC_FRAME_OF_REFERENCE)):
            lcol, rcol = build_column(commit, None, chunk, lenc), build_column(receipt, receipt_nulls, chunk, renc)
            ldev, rdev = DeviceColumn(lcol), DeviceColumn(rcol)
            for condition in CONDITIONS[:6]:
                assert_scan_equal(table_scan_columns(ldev, rdev, condition), oracle_scan_columns(lcol, rcol, condition), f"dates chunk {chunk} lenc {lenc} renc {renc} cond {condition}")


def test_excluded_chunks(device):
    rng = np.random.default_rng(17)
    values = rng.integers(0, 100, 50_000).astype(np.int32)
    host = build_column(values, None, 5000, abi.ENC_DICTIONARY)
    dev = DeviceColumn(host)
    p = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 50)
    full = oracle_scan(host, p)
    got = table_scan(dev, p, excluded_chunks=[1, 4, 9])
    for c in range(host.n_chunks):
        if c in (1, 4, 9):
            assert got.counts[c] == 0
        else:
            np.testing.assert_array_equal(got.pos_list(c), full.pos_list(c))


def test_capacity_error(device):
    values = np.arange(10_000, dtype=np.int32)
    dev = DeviceColumn(build_column(values, None, 1000, abi.ENC_UNENCODED))
    with pytest.raises(abi.HyriseAmdError) as err:
        table_scan(dev, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 5000), capacity=100)
    assert err.value.status == abi.ERR_CAPACITY


def test_type_mismatch_is_rejected(device):
    dev = DeviceColumn(build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_UNENCODED))
    with pytest.raises(abi.HyriseAmdError) as err:
        table_scan(dev, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, 3.5))
    assert err.value.status == abi.ERR_INVALID


def test_large_scan_lineitem_shape(device):
    """l_shipdate-shaped column at 1/10 of SF10 (6M rows, 92 chunks, u16 attribute vectors): oracle parity plus
    size-independent properties (a predicate and its complement partition the rows; results are ascending)."""
    rng = np.random.default_rng(42)
    n = 6_000_000
    days = (rng.integers(0, 2406, n) + rng.integers(1, 122, n)).astype(np.int32)
    host = build_column(days, None, abi.CHUNK_DEFAULT_SIZE, abi.ENC_DICTIONARY)
    assert host.segments[0].width == 2
    dev = DeviceColumn(host)
    total = 0
    for condition, v, v2 in ((abi.PRED_LESS_THAN_EQUALS, 2436, None), (abi.PRED_GREATER_THAN, 2436, None),
                             (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, 731, 1096), (abi.PRED_EQUALS, 1263, None)):
        got = check(host, make_predicate(condition, abi.TYPE_INT, v, v2), dev)
        if condition in (abi.PRED_LESS_THAN_EQUALS, abi.PRED_GREATER_THAN):
            total += sum(int(c) for c in got.counts)
        rows = got.matches[:got.total].astype(np.int64)
        key = rows[:, 0] * (1 << 32) + rows[:, 1]
        assert np.all(np.diff(key) > 0)
    assert total == n


def test_scan_hyrise_binary_tables(device):
    """Columns exactly as Hyrise wrote them (tests/golden/bin: its own DictionarySegment / FrameOfReferenceSegment /
    ValueSegment bytes, parsed by hyrise_amd/binary.py) go to the device unchanged and scan like the oracle says."""
    import glob
    import os
    from hyrise_amd import binary
    from support import GOLDEN
    root = os.path.join(os.path.dirname(GOLDEN), "bin")
    scanned = 0
    for path in sorted(glob.glob(os.path.join(root, "**", "*.bin"), recursive=True)):
        try:
            table = binary.read_table(path)
        except binary.UnsupportedSegment:
            continue
        for c, data_type in enumerate(table.types):
            if data_type == abi.TYPE_STRING or table.chunk_count == 0:
                continue
            values, nulls = binary.decode_column(table, c)
            literal = values[len(values) // 2].item() if len(values) else 0
            dev = DeviceColumn(table.columns[c])
            for condition in (abi.PRED_EQUALS, abi.PRED_NOT_EQUALS, abi.PRED_LESS_THAN, abi.PRED_GREATER_THAN_EQUALS, abi.PRED_IS_NULL, abi.PRED_IS_NOT_NULL):
                p = make_predicate(condition, data_type, literal, nullable=table.nullable[c])
                got = check(table.columns[c], p, dev, context=f"{os.path.relpath(path, root)} column {c} cond {condition}")
                if condition == abi.PRED_LESS_THAN:
                    assert len(result_rows(got)) == int(((values < literal) & ~nulls).sum())   # (ALL_MATCH chunks own no RowIDs)
                scanned += 1
            dev.close()
    assert scanned > 300


def test_lz4_segments_are_decompressed_on_the_device(device):
    """LZ4Segments exactly as Hyrise wrote them (tests/golden/bin/*/LZ4.bin, LZ4MultipleBlocks.bin: blocks compressed one by one against a
    zstd-trained dictionary) cross the C ABI COMPRESSED (HY_ENC_LZ4 + hy_lz4_blocks) and are decompressed by the library's kernel: the
    column reads back as the values of the file's Unencoded twin (hyrise_amd/binary.py's decoder, itself pinned by those files) and scans
    like a ValueSegment.  Synthetic blocks add what the fixtures lack: literal runs with 255-continued lengths, matches that overlap their
    own output, a match that begins in the dictionary and ends in the block; a corrupt block is refused."""
    import glob
    import os
    from hyrise_amd import binary
    from hyrise_amd.storage import HostColumn, HostSegment
    from support import GOLDEN
    lib = device
    root = os.path.join(os.path.dirname(GOLDEN), "bin")
    checked = 0
    for path in sorted(glob.glob(os.path.join(root, "**", "LZ4*.bin"), recursive=True)):
        table = binary.read_table(path, keep_lz4=True)
        for c, data_type in enumerate(table.types):
            if data_type == abi.TYPE_STRING or table.chunk_count == 0:
                continue
            column = table.columns[c]
            if not any(segment.encoding == abi.ENC_LZ4 for segment in column.segments):
                continue
            dev = DeviceColumn(column)
            for chunk, segment in enumerate(column.segments):
                if segment.encoding != abi.ENC_LZ4:
                    continue
                got = np.zeros(segment.size, dtype=segment.decoded.dtype)
                null_words = np.zeros((segment.size + 63) // 64 + 1, dtype=np.uint64)
                abi.check(lib.hy_column_read_chunk(dev.handle, chunk, got.ctypes.data, null_words.ctypes.data))
                nulls = table.null_masks[c][chunk]
                keep = slice(None) if nulls is None else ~nulls
                assert got[keep].tobytes() == segment.decoded[keep].tobytes(), f"{os.path.relpath(path, root)} column {c} chunk {chunk}"
                checked += 1
            dev.close()
    assert checked >= 20

    def encode(literal_runs_and_matches, dictionary=b""):   # a hand-made LZ4 block: [(literals, match offset or None, match length)]
        out = bytearray()
        for literals, offset, length in literal_runs_and_matches:
            token_literals, token_match = min(len(literals), 15), 0 if offset is None else min(length - 4, 15)
            out.append(token_literals << 4 | token_match)
            rest = len(literals) - 15
            while len(literals) >= 15 and rest >= 0:
                out.append(min(rest, 255))
                if rest < 255:
                    break
                rest -= 255
            out += literals
            if offset is None:
                break
            out += bytes([offset & 0xFF, offset >> 8])
            rest = length - 4 - 15
            while length - 4 >= 15 and rest >= 0:
                out.append(min(rest, 255))
                if rest < 255:
                    break
                rest -= 255
        return bytes(out)

    dictionary = bytes(range(200)) * 6                                                # 1200 bytes of history
    sequences = [(bytes(range(7)) * 100, 1, 900),                                       # 700 literals (continued length), then the last byte 900 times
                 (b"", 1200 + 1600 - 100, 300),                                         # 300 bytes out of the dictionary (it lies 1600 output bytes back, the match starts 100 bytes into it)
                 (b"xy", 2, 2000), (b"tail" * 25 + b"zz", None, 0)]                      # "xyxyxy...": a match that overlaps its own output; the last sequence: literals only
    block = encode(sequences, dictionary)
    expected = binary.lz4_block_decode(block, 700 + 900 + 300 + 2 + 2000 + 102, dictionary)
    assert len(expected) % 4 == 0 and expected[1600:1604] == dictionary[100:104]
    values = np.frombuffer(expected, dtype=np.int32)
    segment = HostSegment(abi.ENC_LZ4, abi.TYPE_INT, len(values), 4, None)
    segment.lz4 = ([block], len(expected), len(expected), dictionary)
    dev = DeviceColumn(HostColumn([segment], abi.TYPE_INT))
    got = np.zeros(len(values), dtype=np.int32)
    abi.check(lib.hy_column_read_chunk(dev.handle, 0, got.ctypes.data, None))
    assert got.tobytes() == expected
    dev.close()
    corrupt = HostSegment(abi.ENC_LZ4, abi.TYPE_INT, len(values), 4, None)
    corrupt.lz4 = ([block[:-7]], len(expected), len(expected), dictionary)
    with pytest.raises(abi.HyriseAmdError):
        DeviceColumn(HostColumn([corrupt], abi.TYPE_INT))


def test_run_length_segments(device):
    """RunLengthSegment<T> columns (expanded once by the residency cache) behave like the ValueSegments they decode to:
    scans, a join and an aggregate over long runs with NULL runs."""
    from hyrise_amd.operators import aggregate_hash, join_hash
    from support import oracle_aggregate, oracle_join
    rng = np.random.default_rng(23)
    n, chunk = 200_000, 30_000
    values = np.repeat(rng.integers(0, 50, n // 40 + 1), 40)[:n].astype(np.int32)
    nulls = np.repeat(rng.random(n // 25 + 1) < 0.1, 25)[:n]
    segments = [storage.encode_run_length(values[b:b + chunk], nulls[b:b + chunk]) for b in range(0, n, chunk)]
    host = storage.HostColumn(segments, abi.TYPE_INT)
    assert segments[0].aux_size < chunk // 10
    dev = DeviceColumn(host)
    for condition in CONDITIONS:
        check(host, make_predicate(condition, abi.TYPE_INT, 10, 30, nullable=True), dev, context=f"run length cond {condition}")
    # runs of 1 .. 6000 rows (a wave's 2048 rows lie in one run, in a few, or in more than the 64 it walks), NULL runs, ragged chunks
    lengths = np.concatenate([rng.integers(1, 6_000, 60), rng.integers(1, 12, 3_000), rng.integers(500, 3_000, 40)])
    mixed = np.repeat(rng.integers(0, 50, len(lengths)), lengths).astype(np.int32)
    mixed_nulls = np.repeat(rng.random(len(lengths)) < 0.15, lengths)
    bounds = [0, 65_535, 65_535 + 2_048, 65_535 + 2_048 + 40_001, len(mixed)]
    mixed_host = storage.HostColumn([storage.encode_run_length(mixed[b:e], mixed_nulls[b:e]) for b, e in zip(bounds[:-1], bounds[1:]) if e > b], abi.TYPE_INT)
    mixed_dev = DeviceColumn(mixed_host)
    for condition in CONDITIONS:
        check(mixed_host, make_predicate(condition, abi.TYPE_INT, 10, 30, nullable=True), mixed_dev, context=f"run length (mixed runs) cond {condition}")
    other_host = build_column(rng.integers(0, 60, 3_000).astype(np.int32), None, 1_000, abi.ENC_DICTIONARY)
    other = DeviceColumn(other_host)
    got, want = join_hash(other, dev, abi.JOIN_INNER), oracle_join(other_host, host, abi.JOIN_INNER)
    assert got.n_pairs == want.n_pairs and got.left[:got.n_pairs].tobytes() == want.left[:want.n_pairs].tobytes()
    assert got.right[:got.n_pairs].tobytes() == want.right[:want.n_pairs].tobytes()
    sums = aggregate_hash([dev], [(abi.AGG_COUNT, None), (abi.AGG_SUM, dev)])
    expected = oracle_aggregate([host], [(abi.AGG_COUNT, None), (abi.AGG_SUM, host)])
    assert sums.n_groups == expected.n_groups and sums.column(0) == expected.column(0) and sums.column(1) == expected.column(1)


def device_pos_list(lib, host_column, device_column, predicate, layout=abi.POSLIST_DENSE):
    """hy_table_scan into device memory + hy_poslist_translate: the PosList the next operator reads, copied back for the check
    (HY_POSLIST_CHUNK_REGIONS: the filled prefixes of the chunk regions, concatenated)."""
    rows, n_chunks = max(1, host_column.rows), host_column.n_chunks
    regions, offsets, counts = DeviceArray(lib, (rows, 2), np.uint32), DeviceArray(lib, (n_chunks + 1,), np.int64), DeviceArray(lib, (max(1, n_chunks),), np.int32)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS | abi.SCAN_MATERIALIZE_ALL_MATCH
    result.matches, result.capacity, result.offsets, result.counts = regions.pointer, rows, offsets.pointer, counts.pointer
    abi.check(lib.hy_table_scan(device_column.handle, C.byref(predicate), None, 0, C.byref(result)))
    out = DeviceArray(lib, (rows, 2), np.uint32)
    written = C.c_uint64(0)
    abi.check(lib.hy_poslist_translate(device_column.handle, C.byref(result), layout, out.pointer, rows, C.byref(written)))
    result._keep = (regions, offsets, counts)   # the device buffers live as long as the struct that points at them
    if layout == abi.POSLIST_CHUNK_REGIONS:
        begin, count, everything = offsets.numpy(), counts.numpy(), out.numpy()
        parts = [everything[int(begin[c]):int(begin[c]) + int(count[c])] for c in range(n_chunks)]
        assert sum(len(part) for part in parts) == written.value
        return (np.concatenate(parts) if parts else everything[:0]), result
    return out.numpy()[:written.value], result


def expected_pos_list(host_column, predicate):
    """What TableScan::_on_execute assembles (table_scan.cpp:158-196): the oracle's matches, chunk after chunk, each replaced by
    the RowID it stands for when the input is a reference table."""
    want = oracle_scan(host_column, predicate, flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
    out = []
    for c, segment in enumerate(host_column.segments):
        matches = want.pos_list(c)
        if segment.encoding != abi.ENC_REFERENCE:
            out.append(matches)
        elif segment.data is None:
            out.append(np.stack([np.full(len(matches), segment.ref_chunk_id, dtype=np.uint32), matches[:, 1]], axis=1))
        else:
            out.append(np.asarray(segment.data, dtype=np.uint32).reshape(-1, 2)[matches[:, 1]])
    return np.concatenate(out) if out else np.zeros((0, 2), dtype=np.uint32)


def test_poslist_translate(device):
    """The device-resident hand-over between operators: data columns (packing only), single-chunk and entire-chunk PosLists,
    PosLists over many chunks with NULL RowIDs, an empty table; and the capacity error."""
    rng = np.random.default_rng(23)
    n, chunk = 150_000, 9_000
    values = rng.integers(0, 1000, n).astype(np.int32)
    nulls = rng.random(n) < 0.05
    for encoding in ENCODINGS:
        base = build_column(values, nulls, chunk, encoding)
        base_dev = DeviceColumn(base)
        first = oracle_scan(base, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 700, nullable=True), flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        single = [first.pos_list(c).copy() for c in range(base.n_chunks)] + [2]
        single_host = storage.make_reference_column(base, single, list(range(base.n_chunks)) + [2])
        many = []
        for size in (0, 1, 65, 30_000, 65_535):
            rows = rng.integers(0, n, size)
            pos = np.stack([rows // chunk, rows % chunk], axis=1).astype(np.uint32)
            pos[rng.random(size) < 0.03] = 0xFFFFFFFF
            many.append(pos)
        many_host = storage.make_reference_column(base, many, [None] * len(many))
        for host, dev in ((base, base_dev), (single_host, DeviceColumn(single_host, refs={id(base): base_dev})),
                          (many_host, DeviceColumn(many_host, refs={id(base): base_dev}))):
            for condition in (abi.PRED_EQUALS, abi.PRED_LESS_THAN, abi.PRED_BETWEEN_INCLUSIVE, abi.PRED_IS_NULL, abi.PRED_IS_NOT_NULL):
                p = make_predicate(condition, abi.TYPE_INT, 100, 400, nullable=True)
                want = expected_pos_list(host, p)
                for layout in (abi.POSLIST_DENSE, abi.POSLIST_CHUNK_REGIONS):
                    got, _ = device_pos_list(device, host, dev, p, layout)
                    assert got.tobytes() == want.tobytes(), f"enc {encoding} cond {condition} layout {layout}"
    # too small an output buffer: the needed capacity comes back with HY_ERR_CAPACITY
    p = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 500, nullable=True)
    full, result = device_pos_list(device, base, base_dev, p)
    small = DeviceArray(device, (16, 2), np.uint32)
    written = C.c_uint64(0)
    assert device.hy_poslist_translate(base_dev.handle, C.byref(result), abi.POSLIST_DENSE, small.pointer, 16, C.byref(written)) == abi.ERR_CAPACITY
    assert written.value == len(full) > 16
    assert device.hy_poslist_translate(base_dev.handle, C.byref(result), abi.POSLIST_CHUNK_REGIONS, small.pointer, 16, C.byref(written)) == abi.ERR_CAPACITY
    assert device.hy_poslist_translate(base_dev.handle, C.byref(result), 7, small.pointer, 16, C.byref(written)) == abi.ERR_INVALID
    # a host-memory result is refused
    host_result = abi.ScanResult()
    host_result.mem = abi.MEM_HOST
    assert device.hy_poslist_translate(base_dev.handle, C.byref(host_result), abi.POSLIST_DENSE, small.pointer, 16, C.byref(written)) == abi.ERR_INVALID


SORT_OF = {"AscendingNullsFirst": abi.SORT_ASCENDING_NULLS_FIRST, "DescendingNullsFirst": abi.SORT_DESCENDING_NULLS_FIRST,
           "AscendingNullsLast": abi.SORT_ASCENDING_NULLS_LAST, "DescendingNullsLast": abi.SORT_DESCENDING_NULLS_LAST}
SEGMENT_KINDS = ["Unencoded", "Dictionary", "FrameOfReference", "RunLength", "BitPackedDictionary", "BitPackedFrameOfReference"]


def encode_chunk(values, nulls, kind, nullable):
    """One chunk in one of the layouts the device reads IN PLACE: RunLength segments and BitPackingVectors stay compressed in device
    memory (the scan unpacks them in registers; csrc/scan.hip load_element / run_of_row)."""
    if kind == "RunLength":
        return storage.encode_run_length(values, nulls)
    base = {"Unencoded": abi.ENC_UNENCODED, "Dictionary": abi.ENC_DICTIONARY, "FrameOfReference": abi.ENC_FRAME_OF_REFERENCE,
            "BitPackedDictionary": abi.ENC_DICTIONARY, "BitPackedFrameOfReference": abi.ENC_FRAME_OF_REFERENCE}[kind]
    segment = build_column(values, nulls, max(1, len(values)), base, nullable=nullable).segments[0] if len(values) else storage.encode_segment(values, nulls, base)
    return storage.bit_pack_segment(segment) if kind.startswith("BitPacked") else segment


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64], ids=lambda t: t.__name__)
@pytest.mark.parametrize("kind", SEGMENT_KINDS)
@pytest.mark.parametrize("sort_mode", KA.BETWEEN_SORT_MODES)
@pytest.mark.parametrize("nullable", [False, True], ids=["not_null", "nullable"])
def test_between_known_answers(device, np_type, kind, sort_mode, nullable):
    """table_scan_between_test.cpp:194-243 on the device, on the ENCODED segments (the first two chunks encoded, the last a ValueSegment,
    :40-96) and with the chunks flagged as sorted where the reference flags them: the reference's expected row lists, and the oracle's
    PosLists."""
    if kind.endswith("FrameOfReference") and np_type != np.int32:
        pytest.skip("encoding_supports_data_type(): FrameOfReference holds int only")
    a, nulls, b = KA.between_table(np_type, sort_mode, nullable)
    plain = build_column(a, nulls, 6, abi.ENC_UNENCODED, nullable=nullable)
    encoded = [encode_chunk(a[c * 6:c * 6 + 6], None if nulls is None else nulls[c * 6:c * 6 + 6], kind, nullable) for c in range(2)]
    host = storage.HostColumn(encoded + plain.segments[2:], plain.data_type)
    if sort_mode != "unsorted":   # (NULLs in front, :75)
        for segment in host.segments:
            segment.sorted_by = abi.SORT_ASCENDING_NULLS_FIRST if sort_mode == "ascending" else abi.SORT_DESCENDING_NULLS_FIRST
    dev = DeviceColumn(host)
    data_type = storage.TYPE_OF_NP[np.dtype(np_type)]
    cast = (lambda x: np_type(int(x))) if np.issubdtype(np_type, np.integer) else np_type
    for condition, tests in KA.BETWEEN_TESTS.items():
        for lower, upper, expected in tests:
            p = make_predicate(condition, data_type, cast(lower), cast(upper), nullable=nullable)
            got = check(host, p, dev, context=f"condition {condition} BETWEEN {lower} AND {upper}")
            rows = np.array([c * 6 + o for c, o in result_rows(got)], dtype=np.int64)
            assert sorted(b[rows].tolist()) == KA.between_expected(expected, sort_mode, nullable)


@pytest.mark.parametrize("sort_mode", KA.SORTED_SEGMENT_SORT_MODES)
@pytest.mark.parametrize("null_usage", KA.SORTED_SEGMENT_NULL_USAGES)
@pytest.mark.parametrize("kind", SEGMENT_KINDS)
@pytest.mark.parametrize("flagged", [True, False], ids=["flagged", "not_flagged"])
def test_sorted_segment_search_known_answers(device, sort_mode, null_usage, kind, flagged):
    """table_scan_sorted_segment_search_test.cpp:106-214 on the device, every segment layout: with the chunk's sort flag the scan finds
    the rows with binary searches (prepare_jobs: JOB_RANGE) and reads nothing, without it every row is tested -- same positions, same
    order, either way."""
    values, nulls = KA.sorted_search_segment(sort_mode, null_usage)
    host = storage.HostColumn([encode_chunk(values, nulls, kind, nulls is not None)], abi.TYPE_INT)
    if flagged:
        host.segments[0].sorted_by = SORT_OF[sort_mode]
    dev = DeviceColumn(host)
    for condition, value, value2, expected in KA.SORTED_SEGMENT_SEARCH_TESTS:
        got = check(host, make_predicate(condition, abi.TYPE_INT, value, value2, nullable=nulls is not None), dev, context=f"condition {condition} value {value} / {value2}")
        rows = [o for _, o in result_rows(got)]
        want = [] if null_usage == "OnlyNulls" else (expected if sort_mode.startswith("Ascending") else expected[::-1])
        assert values[rows].tolist() == want


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64], ids=lambda t: t.__name__)
@pytest.mark.parametrize("sort_mode", KA.SORTED_SEGMENT_SORT_MODES)
def test_sorted_chunks_of_full_size(device, np_type, sort_mode):
    """Chunks of 65 535 rows, each sorted on its own and flagged (Chunk::individually_sorted_by): the 64-ary searches of prepare_jobs
    against the oracle's full scan -- every condition, literals below / at / between / above the chunk's values, NULL blocks of
    different lengths (none, some, a whole chunk), every layout, through the streaming and the generic kernel."""
    rng = np.random.default_rng(77)
    ascending, nulls_last = sort_mode.startswith("Ascending"), sort_mode.endswith("NullsLast")
    data_type = storage.TYPE_OF_NP[np.dtype(np_type)]
    sizes = [65535, 65535, 40_000, 1, 64, 65]
    null_counts = [0, 1000, 40_000, 0, 64, 1]
    chunks = []
    for size, n_null in zip(sizes, null_counts):
        body = np.sort(rng.integers(-500, 500, size - n_null) * 2).astype(np_type)   # even values: odd literals fall between them
        if not ascending:
            body = body[::-1]
        filler = np.zeros(n_null, np_type)
        values = np.concatenate([body, filler] if nulls_last else [filler, body])
        mask = np.concatenate([np.zeros(len(body), bool), np.ones(n_null, bool)] if nulls_last else [np.ones(n_null, bool), np.zeros(len(body), bool)])
        chunks.append((values, mask))
    kinds = [k for k in SEGMENT_KINDS if np_type == np.int32 or not k.endswith("FrameOfReference")]
    for kind in kinds:
        host = storage.HostColumn([encode_chunk(v, m, kind, True) for v, m in chunks], data_type)
        for segment in host.segments:
            segment.sorted_by = SORT_OF[sort_mode]
        dev = DeviceColumn(host)
        for condition in CONDITIONS[:10]:
            for value, value2 in ((-2000, 2000), (-1000, 998), (0, 0), (-301, 299), (300, -300), (7, 7), (999, 1001), (-1002, -1001), (-600, 601)):
                for flags in (0, abi.SCAN_MATERIALIZE_ALL_MATCH):
                    p = make_predicate(condition, data_type, np_type(value), np_type(value2), nullable=True)
                    check(host, p, dev, flags, context=f"sorted {sort_mode} {kind} {np_type.__name__} cond {condition} lit {value},{value2} flags {flags}")


def test_compressed_segments_through_every_operator(device):
    """RunLength segments and BitPackingVectors stay compressed in device memory; a scan reads them in place, every other operator reads
    the twin a device kernel decodes once (runtime.hip plain_column): a join, an aggregate, a projection, a ColumnVsColumn scan and a
    scan through reference segments over such columns answer like the oracle over the decoded columns."""
    from hyrise_amd.operators import aggregate_hash, join_hash
    from support import oracle_aggregate, oracle_join
    rng = np.random.default_rng(29)
    n, chunk = 150_000, 40_000
    keys = np.repeat(rng.integers(0, 300, n // 20 + 1), 20)[:n].astype(np.int32)
    nulls = np.repeat(rng.random(n // 50 + 1) < 0.05, 50)[:n]
    measures = rng.integers(0, 5000, n).astype(np.int32)
    for kind in ("RunLength", "BitPackedDictionary", "BitPackedFrameOfReference"):
        host = storage.HostColumn([encode_chunk(keys[b:b + chunk], nulls[b:b + chunk], kind, True) for b in range(0, n, chunk)], abi.TYPE_INT)
        packed = storage.HostColumn([encode_chunk(measures[b:b + chunk], None, "BitPackedDictionary", False) for b in range(0, n, chunk)], abi.TYPE_INT)
        assert kind == "RunLength" or (host.segments[0].width == 0 and 1 <= host.segments[0].bits <= 16)
        dev, packed_dev = DeviceColumn(host), DeviceColumn(packed)
        for condition in CONDITIONS:
            check(host, make_predicate(condition, abi.TYPE_INT, 100, 200, nullable=True), dev, context=f"{kind} cond {condition}")
        other_host = build_column(rng.integers(0, 400, 2_000).astype(np.int32), None, 700, abi.ENC_DICTIONARY)
        other = DeviceColumn(other_host)
        got, want = join_hash(other, dev, abi.JOIN_INNER), oracle_join(other_host, host, abi.JOIN_INNER)
        assert got.n_pairs == want.n_pairs and got.left[:got.n_pairs].tobytes() == want.left[:want.n_pairs].tobytes()
        assert got.right[:got.n_pairs].tobytes() == want.right[:want.n_pairs].tobytes()
        spec = [(abi.AGG_COUNT, None), (abi.AGG_SUM, packed_dev), (abi.AGG_MAX, packed_dev)]
        sums = aggregate_hash([dev], spec)
        expected = oracle_aggregate([host], [(abi.AGG_COUNT, None), (abi.AGG_SUM, packed), (abi.AGG_MAX, packed)])
        assert sums.n_groups == expected.n_groups
        for column in range(3):
            assert sums.column(column) == expected.column(column), f"{kind}: aggregate column {column}"
        got_columns = table_scan_columns(dev, packed_dev, abi.PRED_LESS_THAN)
        assert_scan_equal(got_columns, oracle_scan_columns(host, packed, abi.PRED_LESS_THAN), f"{kind}: ColumnVsColumn")
        # a reference table over the compressed column (an EntireChunkPosList per chunk), scanned
        reference_host = storage.make_reference_column(host, list(range(host.n_chunks)))
        reference = DeviceColumn(reference_host, refs={id(host): dev})
        p = make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_INT, 50, 250, nullable=True)
        assert_scan_equal(table_scan(reference, p), oracle_scan(reference_host, p), f"{kind}: reference segments")


@pytest.mark.parametrize("bits", [1, 2, 3, 5, 7, 8, 9, 11, 12, 13, 15, 16])
def test_bit_packed_vectors_stream(device, bits):
    """A column whose chunks all hold BitPackingVectors of at most 16 bits takes the streaming instantiation scan_slices<16>: eight elements
    are unpacked from the 20 bytes around the group's `bits` bytes (any byte alignment), dictionary value ids with NULLs and FrameOfReference
    offsets, ragged chunks, every condition -- PosLists equal to the oracle's over the unpacked column."""
    rng = np.random.default_rng(1000 + bits)
    chunk = 20_000 if bits <= 13 else 65_535                # (a chunk holds all 2^bits - 1 values)
    n = 3 * chunk + chunk // 2 + 3
    distinct = max(1, (1 << bits) - 1)                      # the NULL value id is `distinct`: it needs exactly `bits` bits
    values = (rng.integers(0, distinct, n) * 3 - 50).astype(np.int32)
    values[:distinct] = np.arange(distinct, dtype=np.int32) * 3 - 50   # (every value occurs in the first chunk at least)
    nulls = rng.random(n) < 0.03
    nulls[:chunk] |= np.arange(chunk) == 17
    for kind, with_nulls in (("BitPackedDictionary", True), ("BitPackedDictionary", False), ("BitPackedFrameOfReference", True)):
        mask = nulls if with_nulls else None
        if kind == "BitPackedFrameOfReference":              # offsets below 2^bits inside every 2048-row block
            bases = (rng.integers(-1000, 1000, n // chunk + 1) * 7)[np.arange(n) // chunk]   # one base per chunk: every 2048-row block of a chunk has offsets below 2^bits
            column_values = (bases + rng.integers(0, 1 << bits, n)).astype(np.int32)
        else:
            column_values = values
        segments = [encode_chunk(column_values[b:b + chunk], None if mask is None else mask[b:b + chunk], kind, with_nulls) for b in range(0, n, chunk)]
        assert all(s.width == 0 and s.bits <= 16 for s in segments)
        if kind == "BitPackedDictionary" and with_nulls:
            assert segments[0].bits == bits
        host = storage.HostColumn(segments, abi.TYPE_INT)
        dev = DeviceColumn(host)
        low, high = int(np.percentile(column_values, 30)), int(np.percentile(column_values, 70))
        for condition in CONDITIONS:
            for value, value2 in ((low, high), (int(column_values[5]), int(column_values[5])), (-10_000, 10_000)):
                for flags in (0, abi.SCAN_MATERIALIZE_ALL_MATCH):
                    p = make_predicate(condition, abi.TYPE_INT, value, value2, nullable=with_nulls)
                    check(host, p, dev, flags, context=f"{kind} {bits} bits nulls {with_nulls} cond {condition} lit {value},{value2} flags {flags}")


def test_upload_windows(device):
    """hy_column_create moves host buffers through 32 MiB windows of pinned memory: a column of many small buffers that fill several
    windows (dictionary segments: three buffers a chunk), and chunks larger than a window (they go straight from the caller's memory) --
    the scans of both equal the oracle's."""
    rng = np.random.default_rng(8)
    many = rng.integers(0, 3000, 24_000_000).astype(np.int32)                # 367 chunks x (attribute vector + dictionary): ~50 MB, two windows
    host = build_column(many, None, 65535, abi.ENC_DICTIONARY)
    check(host, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 1200), context="many buffers")
    big = rng.integers(-1000, 1000, 21_000_000).astype(np.int32)             # two chunks of 42 MB each: larger than a window
    nulls = rng.random(len(big)) < 0.01
    host = build_column(big, nulls, 10_500_000, abi.ENC_UNENCODED)
    check(host, make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_INT, -5, 700, nullable=True), context="chunks larger than a window")
