"""LIKE family on the device: the host evaluates the pattern over each chunk's dictionary (hyrise_amd/like.py), the scan
kernel tests every row's value id against that bitmap (KIND_VALUE_ID_SET) -- ColumnLikeTableScanImpl::_scan_dictionary_segment
(column_like_table_scan_impl.cpp:74-140).  Checked against the reference's expected tables (table_scan_string_test.cpp) and
against the oracle on large random string columns."""
import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.like import dictionary_matches
from hyrise_amd.operators import make_predicate, table_scan
from hyrise_amd.storage import DeviceColumn
from like_cases import SPECIAL_CHARS_CASES, STRING_TABLE_CASES, StringTable, expected_rows
from support import assert_scan_equal, build_column, oracle_scan, result_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("line,condition,pattern,expected", STRING_TABLE_CASES)
def test_like_on_dictionary_segments(device, line, condition, pattern, expected):
    table = StringTable("int_string_like.tbl", 5)
    dev = DeviceColumn(table.column)
    predicate = table.predicate(condition, pattern)
    got = table_scan(dev, predicate)
    assert_scan_equal(got, oracle_scan(table.column, predicate), f"table_scan_string_test.cpp:{line}")
    assert table.rows_of(result_rows(got)) == expected_rows(expected), f"table_scan_string_test.cpp:{line}"


@pytest.mark.parametrize("line,condition,pattern,expected", SPECIAL_CHARS_CASES)
def test_like_special_characters(device, line, condition, pattern, expected):
    table = StringTable("int_string_like_special_chars.tbl", 2)
    got = table_scan(DeviceColumn(table.column), table.predicate(condition, pattern))
    assert table.rows_of(result_rows(got)) == expected_rows(expected), f"table_scan_string_test.cpp:{line}"


@pytest.mark.parametrize("line,condition,pattern,expected", [c for c in STRING_TABLE_CASES if c[0] in (151, 178, 242, 265)])
def test_like_on_referenced_dictionary_segments(device, line, condition, pattern, expected):
    """ScanLike*OnReferencedDictSegment (:156-163,183-190,247-254,269-275)."""
    table = StringTable("int_string_like.tbl", 5)
    a = build_column(table.tbl.columns[0], None, 5, abi.ENC_UNENCODED)
    first = table_scan(DeviceColumn(a), make_predicate(abi.PRED_GREATER_THAN, abi.TYPE_INT, 0), flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
    pos_lists = [first.pos_list(c).copy() for c in range(a.n_chunks)]
    referencing = storage.make_reference_column(table.column, pos_lists, list(range(a.n_chunks)))
    base_dev = DeviceColumn(table.column)
    ref_dev = DeviceColumn(referencing, refs={id(table.column): base_dev})
    predicate = table.predicate(condition, pattern)
    got = table_scan(ref_dev, predicate)
    assert_scan_equal(got, oracle_scan(referencing, predicate), f"table_scan_string_test.cpp:{line}")
    data_rows = [tuple(pos_lists[chunk][offset]) for chunk, offset in result_rows(got)]
    assert table.rows_of(data_rows) == expected_rows(expected)


WORDS = [b"BRASS", b"COPPER", b"NICKEL", b"STEEL", b"TIN", b"PROMO", b"ECONOMY", b"STANDARD", b"ANODIZED", b"BURNISHED", b"PLATED",
         b"POLISHED", b"BRUSHED", b"SMALL", b"MEDIUM", b"LARGE"]


def random_string_column(rng, rows, chunk_size, distinct, null_fraction):
    """p_type-like strings ("STANDARD POLISHED TIN"): `distinct` of them per chunk, so that the attribute vectors come out
    in every width (u8 / u16 / u32 when distinct > 65535)."""
    segments, dictionaries = [], []
    for begin in range(0, rows, chunk_size):
        n = min(chunk_size, rows - begin)
        pool = [b" ".join((WORDS[rng.integers(len(WORDS))], WORDS[rng.integers(len(WORDS))], b"%06d" % i)) for i in range(distinct)]
        values = [pool[i] for i in rng.integers(0, distinct, n)]
        nulls = rng.random(n) < null_fraction if null_fraction else None
        segment, dictionary = storage.encode_string_dictionary(values, nulls)
        segments.append(segment)
        dictionaries.append(dictionary)
    return storage.HostColumn(segments, abi.TYPE_STRING), dictionaries


@pytest.mark.parametrize("distinct,chunk_size", [(40, 20_000), (3_000, 20_000), (70_000, 150_000)])
@pytest.mark.parametrize("null_fraction", [0.0, 0.1])
def test_like_random_string_columns(device, distinct, chunk_size, null_fraction):
    rng = np.random.default_rng(distinct)
    host, dictionaries = random_string_column(rng, 4 * chunk_size + 1234, chunk_size, distinct, null_fraction)
    dev = DeviceColumn(host)
    widths = {s.width for s in host.segments}
    assert widths <= {1, 2, 4}
    for condition, pattern in ((abi.PRED_LIKE, "%BRASS%"), (abi.PRED_NOT_LIKE, "PROMO%"), (abi.PRED_LIKE, "%0_1"), (abi.PRED_LIKE, "%"),
                               (abi.PRED_NOT_LIKE, "%"), (abi.PRED_LIKE_INSENSITIVE, "%steel %"), (abi.PRED_LIKE, "%POLISHED%TIN%"),
                               (abi.PRED_LIKE, "%nothing%")):
        predicate = make_predicate(condition, abi.TYPE_STRING, nullable=null_fraction > 0,
                                   dictionary_matches=dictionary_matches(dictionaries, pattern, condition))
        for flags in (0, abi.SCAN_CHUNK_REGIONS):
            got = table_scan(dev, predicate, flags=flags)
            assert_scan_equal(got, oracle_scan(host, predicate, flags=flags), f"distinct {distinct} pattern {pattern} cond {condition} flags {flags}")


def test_like_rejections(device):
    """LIKE needs a string dictionary column and the host's bitmaps (column_like_table_scan_impl.cpp:32-36: 'LIKE operator
    only applicable on string columns')."""
    ints = build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_DICTIONARY)
    with pytest.raises(Exception):
        table_scan(DeviceColumn(ints), make_predicate(abi.PRED_LIKE, abi.TYPE_STRING))
    table = StringTable("int_string_like.tbl", 5)
    with pytest.raises(Exception):
        table_scan(DeviceColumn(table.column), make_predicate(abi.PRED_LIKE, abi.TYPE_STRING))
