"""Parity of hy_projection_arithmetic with the CPU restatement of the ExpressionEvaluator's arithmetic: bit-identical
values (the float operations are single IEEE operations in the type the reference computes in) and NULLs, for every
operator x operand type pair x encoding, literals on either side, reference-segment inputs; and TPC-H Q6 end to end --
three scans, the product, the sum -- without leaving the device."""
import numpy as np
import pytest

from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import aggregate_hash, make_predicate, projection_arithmetic, table_scan
from hyrise_amd.storage import DeviceColumn
from support import build_column, load_tbl, oracle_arithmetic

pytestmark = pytest.mark.gpu
OPS = [abi.ARITH_ADD, abi.ARITH_SUB, abi.ARITH_MUL, abi.ARITH_DIV, abi.ARITH_MOD]
NP = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}


def assert_same(got, want, context):
    values, nulls = got.read()
    want_values, want_nulls = want
    assert values.dtype == want_values.dtype, context
    np.testing.assert_array_equal(nulls, want_nulls, err_msg=f"NULLs {context}")
    keep = ~nulls
    assert values[keep].tobytes() == want_values[keep].tobytes(), f"values {context}"
    assert not values[nulls].any(), f"NULL cells hold T{{}} {context}"


def test_reference_series_on_device(device):   # expression_evaluator_to_values_test.cpp:244-256
    t = load_tbl("expression_evaluator/input_a.tbl")
    cols = {name: build_column(*t.column(name), 3, abi.ENC_UNENCODED) for name in "abc"}
    dev = {name: DeviceColumn(c) for name, c in cols.items()}
    values, nulls = projection_arithmetic(abi.ARITH_MUL, dev["a"], dev["b"]).read()
    assert values.tolist() == [2, 6, 12, 20] and not nulls.any()
    values, nulls = projection_arithmetic(abi.ARITH_MOD, dev["a"], dev["c"]).read()
    assert [None if n else v for v, n in zip(values.tolist(), nulls)] == [1, None, 3, None]
    inner = projection_arithmetic(abi.ARITH_ADD, dev["b"], dev["c"])           # a + (b + c): a result column as an operand
    values, nulls = projection_arithmetic(abi.ARITH_ADD, dev["a"], inner).read()
    assert [None if n else v for v, n in zip(values.tolist(), nulls)] == [36, None, 41, None]
    values, nulls = projection_arithmetic(abi.ARITH_ADD, dev["a"], None).read()
    assert nulls.all()


def test_all_type_pairs(device):
    rng = np.random.default_rng(61)
    n, chunk = 40_000, 9_000
    raw = {abi.TYPE_INT: rng.integers(-50, 50, n).astype(np.int32), abi.TYPE_LONG: (rng.integers(-5, 5, n) * 3_000_000_000).astype(np.int64),
           abi.TYPE_FLOAT: (rng.integers(-400, 400, n) / 7.0).astype(np.float32), abi.TYPE_DOUBLE: rng.normal(0, 1000, n)}
    raw[abi.TYPE_INT][::11] = 0                      # zero divisors
    raw[abi.TYPE_FLOAT][::13] = 0.0
    raw[abi.TYPE_INT][5] = np.iinfo(np.int32).max    # int32 wrap-around
    nulls = {t: (rng.random(n) < 0.1) for t in raw}
    hosts, devs = {}, {}
    for t, values in raw.items():
        encoding = abi.ENC_DICTIONARY if t in (abi.TYPE_INT, abi.TYPE_FLOAT) else abi.ENC_UNENCODED
        hosts[t] = build_column(values, nulls[t], chunk, encoding)
        devs[t] = DeviceColumn(hosts[t])
    literal = {abi.TYPE_INT: 3, abi.TYPE_LONG: 5_000_000_000, abi.TYPE_FLOAT: 2.5, abi.TYPE_DOUBLE: 0.1}
    for op in OPS:
        for lt in raw:
            for rt in raw:
                got = projection_arithmetic(op, devs[lt], devs[rt])
                assert_same(got, oracle_arithmetic(op, (raw[lt], nulls[lt]), (raw[rt], nulls[rt])), f"op {op} types {lt},{rt}")
            for rt, value in literal.items():
                assert_same(projection_arithmetic(op, devs[lt], (rt, value)), oracle_arithmetic(op, (raw[lt], nulls[lt]), (rt, value)), f"op {op} {lt} x literal {rt}")
                assert_same(projection_arithmetic(op, (rt, value), devs[lt]), oracle_arithmetic(op, (rt, value), (raw[lt], nulls[lt])), f"op {op} literal {rt} x {lt}")
    # FrameOfReference input and reference segments (a scan's output, gathered)
    for_host = build_column(raw[abi.TYPE_INT], None, chunk, abi.ENC_FRAME_OF_REFERENCE)
    for_dev = DeviceColumn(for_host)
    assert_same(projection_arithmetic(abi.ARITH_MUL, for_dev, devs[abi.TYPE_DOUBLE]),
                oracle_arithmetic(abi.ARITH_MUL, (raw[abi.TYPE_INT], None), (raw[abi.TYPE_DOUBLE], nulls[abi.TYPE_DOUBLE])), "FrameOfReference x double")
    rows = rng.integers(0, n, 12_345)
    pos = np.stack([rows // chunk, rows % chunk], axis=1).astype(np.uint32)
    refs = []
    for t in (abi.TYPE_FLOAT, abi.TYPE_LONG):
        ref_host = storage.make_reference_column(hosts[t], [pos], [None])
        refs.append(DeviceColumn(ref_host, refs={id(hosts[t]): devs[t]}))
    assert_same(projection_arithmetic(abi.ARITH_SUB, refs[0], refs[1]),
                oracle_arithmetic(abi.ARITH_SUB, (raw[abi.TYPE_FLOAT][rows], nulls[abi.TYPE_FLOAT][rows]), (raw[abi.TYPE_LONG][rows], nulls[abi.TYPE_LONG][rows])),
                "reference segments float - long")


def test_tpch_q6_pipeline_stays_on_device(device, monkeypatch):
    """SELECT SUM(l_extendedprice * l_discount) FROM lineitem WHERE l_shipdate >= 1994-01-01 AND l_shipdate < 1995-01-01
    AND l_discount BETWEEN 0.05 AND 0.07 AND l_quantity < 24 -- tpch.run_q6: three scans chained through reference segments,
    the product over the survivors, the sum.  Every intermediate is a device buffer: the scans write chunk regions to HBM,
    hy_poslist_translate packs / dereferences them there, the next operator reads them in place.  Proof: the host-result
    classes are made unusable for the duration of the chain, and every PosList handed on is a CUDA tensor."""
    import torch
    from hyrise_amd import operators
    from hyrise_amd.distributed import HipExecutor
    data = tpch.TpchData(scale_factor=0.05, seed=7)
    columns = {name: DeviceColumn(column) for name, column in tpch.q6_columns(data, chunk_size=20_000).items()}
    ex = HipExecutor(torch.device("cuda", 0))
    handed_on = []
    real_reference_column = ex.reference_column_chunked

    def reference_column(base, pos_lists):
        handed_on.append(pos_lists)
        return real_reference_column(base, pos_lists)

    def forbidden(*args, **kwargs):
        raise AssertionError("a PosList was read on the host")

    monkeypatch.setattr(ex, "reference_column_chunked", reference_column)
    monkeypatch.setattr(operators.HostScanResult, "__init__", forbidden)      # no host-memory scan result may even be created
    monkeypatch.setattr(operators.HostScanResult, "pos_list", forbidden)
    revenue, qualifying = tpch.run_q6(ex, columns)
    monkeypatch.undo()
    assert len(handed_on) == 4 and all(lists.rows.is_cuda and lists.rows.dtype == torch.int32 for lists in handed_on)
    keep = (data.l_shipdate >= tpch.DAY_1994_01_01) & (data.l_shipdate < tpch.DAY_1995_01_01) & (data.l_discount >= np.float32(0.05)) & \
           (data.l_discount <= np.float32(0.07)) & (data.l_quantity < 24)
    # the last PosLists are exactly the qualifying rows, chunk by chunk in table order, as RowIDs of the DATA table -- and every
    # list still references the one chunk it came from
    last = handed_on[-1]
    everything = last.rows.cpu().numpy().view(np.uint32)
    final = np.concatenate([everything[int(b):int(b) + int(n)] for b, n in zip(last.begin, last.count)])
    np.testing.assert_array_equal(final[:, 0].astype(np.int64) * 20_000 + final[:, 1], np.flatnonzero(keep))
    assert last.base_chunk.tolist() == list(range(len(last.count)))
    for c, (b, n) in enumerate(zip(last.begin, last.count)):
        assert (everything[int(b):int(b) + int(n), 0] == c).all()
    products = data.l_extendedprice[keep] * data.l_discount[keep]            # float32 products, like the reference
    assert qualifying == int(keep.sum()) > 100
    assert abs(revenue - float(products.astype(np.float64).sum())) <= 1e-9 * abs(float(products.astype(np.float64).sum()))


def test_plain_operands_all_type_pairs(device):
    """The fast path of projection_rows: + - * over unencoded value segments without NULLs and literals (wide loads, the
    cell arithmetic instantiated per type pair), chunk sizes that are no multiple of four rows, every type pair and literal
    side -- bit for bit what the oracle computes, no NULL in the result."""
    rng = np.random.default_rng(62)
    n, chunk = 40_003, 9_001
    raw = {abi.TYPE_INT: rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), abi.TYPE_LONG: rng.integers(-2**62, 2**62, n).astype(np.int64),
           abi.TYPE_FLOAT: (rng.normal(0, 1e6, n)).astype(np.float32), abi.TYPE_DOUBLE: rng.normal(0, 1e12, n)}
    devs = {t: DeviceColumn(build_column(values, None, chunk, abi.ENC_UNENCODED)) for t, values in raw.items()}
    literal = {abi.TYPE_INT: -7, abi.TYPE_LONG: 5_000_000_000, abi.TYPE_FLOAT: 0.3, abi.TYPE_DOUBLE: 1.0 - 0.05}
    for op in (abi.ARITH_ADD, abi.ARITH_SUB, abi.ARITH_MUL):
        for lt in raw:
            for rt in raw:
                got = projection_arithmetic(op, devs[lt], devs[rt])
                assert_same(got, oracle_arithmetic(op, (raw[lt], None), (raw[rt], None)), f"plain op {op} types {lt},{rt}")
            for rt, value in literal.items():
                assert_same(projection_arithmetic(op, devs[lt], (rt, value)), oracle_arithmetic(op, (raw[lt], None), (rt, value)), f"plain op {op} {lt} x literal {rt}")
                assert_same(projection_arithmetic(op, (rt, value), devs[lt]), oracle_arithmetic(op, (rt, value), (raw[lt], None)), f"plain op {op} literal {rt} x {lt}")
    # a result column (pooled device buffers) as an operand of the next projection: l_extendedprice * (1 - l_discount)
    one_minus = projection_arithmetic(abi.ARITH_SUB, (abi.TYPE_INT, 1), devs[abi.TYPE_FLOAT])
    chained = projection_arithmetic(abi.ARITH_MUL, devs[abi.TYPE_DOUBLE], one_minus)
    inner_values, _ = oracle_arithmetic(abi.ARITH_SUB, (abi.TYPE_INT, 1), (raw[abi.TYPE_FLOAT], None))
    assert_same(chained, oracle_arithmetic(abi.ARITH_MUL, (raw[abi.TYPE_DOUBLE], None), (inner_values, None)), "chained plain projections")


def test_plain_operands_with_null_bitmaps(device):
    """The same fast path over unencoded value segments that carry a null bitmap (what an earlier projection leaves, always): a cell is NULL
    where an operand's is and holds T{}; chunk sizes that are no multiple of four rows (the group that straddles a chunk's end is read row
    by row), every type pair, a NULL-free partner and a literal -- bit for bit the oracle's."""
    rng = np.random.default_rng(63)
    n, chunk = 40_003, 9_001
    raw = {abi.TYPE_INT: rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), abi.TYPE_LONG: rng.integers(-2**62, 2**62, n).astype(np.int64),
           abi.TYPE_FLOAT: (rng.normal(0, 1e6, n)).astype(np.float32), abi.TYPE_DOUBLE: rng.normal(0, 1e12, n)}
    nulls = {t: rng.random(n) < 0.07 for t in raw}
    nulls[abi.TYPE_INT][-3:] = [True, False, True]   # (in the straddling group)
    with_nulls = {t: DeviceColumn(build_column(values, nulls[t], chunk, abi.ENC_UNENCODED)) for t, values in raw.items()}
    without = {t: DeviceColumn(build_column(values, None, chunk, abi.ENC_UNENCODED)) for t, values in raw.items()}
    for op in (abi.ARITH_ADD, abi.ARITH_SUB, abi.ARITH_MUL):
        for lt in raw:
            for rt in raw:
                assert_same(projection_arithmetic(op, with_nulls[lt], with_nulls[rt]), oracle_arithmetic(op, (raw[lt], nulls[lt]), (raw[rt], nulls[rt])), f"op {op} types {lt},{rt}, both nullable")
                assert_same(projection_arithmetic(op, without[lt], with_nulls[rt]), oracle_arithmetic(op, (raw[lt], None), (raw[rt], nulls[rt])), f"op {op} types {lt},{rt}, right nullable")
            assert_same(projection_arithmetic(op, with_nulls[lt], (abi.TYPE_INT, 3)), oracle_arithmetic(op, (raw[lt], nulls[lt]), (abi.TYPE_INT, 3)), f"op {op} {lt} nullable x literal")
    # results as operands: (1 - x) * (1 + y), NULL where x or y is
    left = projection_arithmetic(abi.ARITH_SUB, (abi.TYPE_INT, 1), with_nulls[abi.TYPE_FLOAT])
    right = projection_arithmetic(abi.ARITH_ADD, (abi.TYPE_INT, 1), with_nulls[abi.TYPE_DOUBLE])
    left_values, left_nulls = oracle_arithmetic(abi.ARITH_SUB, (abi.TYPE_INT, 1), (raw[abi.TYPE_FLOAT], nulls[abi.TYPE_FLOAT]))
    right_values, right_nulls = oracle_arithmetic(abi.ARITH_ADD, (abi.TYPE_INT, 1), (raw[abi.TYPE_DOUBLE], nulls[abi.TYPE_DOUBLE]))
    assert_same(projection_arithmetic(abi.ARITH_MUL, left, right), oracle_arithmetic(abi.ARITH_MUL, (left_values, left_nulls), (right_values, right_nulls)), "chained nullable projections")


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64], ids=lambda t: t.__name__)
def test_export_through_pos_lists(device, np_type):
    """hy_column_export of a reference column (the foreign key of a join result, materialised for the next join): PosLists over several
    chunks in random order with NULL RowIDs, and an EntireChunk / single-chunk PosList per chunk (scan output), over Unencoded /
    Dictionary / FrameOfReference base segments with NULLs -- the values and NULL bytes of the referenced cells."""
    import torch
    from hyrise_amd.distributed import DevicePosLists, HipExecutor
    rng = np.random.default_rng(5)
    n, chunk = 200_000, 30_000
    values = (rng.integers(-5000, 5000, n) if np.issubdtype(np_type, np.integer) else rng.random(n) * 1000 - 500).astype(np_type)
    nulls = rng.random(n) < 0.05
    encodings = [abi.ENC_UNENCODED, abi.ENC_DICTIONARY, abi.ENC_FRAME_OF_REFERENCE, abi.ENC_DICTIONARY, abi.ENC_UNENCODED, abi.ENC_FRAME_OF_REFERENCE, abi.ENC_DICTIONARY]
    ex = HipExecutor(torch.device("cuda", 0))
    for with_nulls in (False, True):
        base = DeviceColumn(build_column(values, nulls if with_nulls else None, chunk, encodings))
        picks = rng.integers(0, n, 500_000)
        rows = np.stack([picks // chunk, picks % chunk], axis=1).astype(np.uint32)
        null_rows = rng.random(len(picks)) < 0.01
        rows[null_rows] = 0xFFFFFFFF                                   # NULL_ROW_ID (outer joins)
        through = ex.reference_column(base, torch.from_numpy(rows.view(np.int32)).to("cuda"), 65535)
        got_values, got_nulls = ex.export(through)
        want_null = null_rows | (nulls[picks] if with_nulls else False)
        np.testing.assert_array_equal(got_nulls.cpu().numpy().astype(bool), want_null)
        want = np.where(want_null, np_type(0), values[picks])
        assert got_values.cpu().numpy().tobytes() == want.astype(np_type).tobytes()
        # one PosList per base chunk (what a scan hands on): every second row of the chunk, the last chunk whole
        counts = [(min(n, (c + 1) * chunk) - c * chunk + 1) // 2 for c in range((n + chunk - 1) // chunk)]
        offsets = np.concatenate([np.arange(0, min(n, (c + 1) * chunk) - c * chunk, 2) for c in range(len(counts))])
        chunk_ids = np.concatenate([np.full(count, c) for c, count in enumerate(counts)])
        lists = np.stack([chunk_ids, offsets], axis=1).astype(np.uint32)
        begin = np.concatenate([[0], np.cumsum(counts)[:-1]])
        pos_lists = DevicePosLists(torch.from_numpy(lists.view(np.int32)).to("cuda"), begin, np.array(counts), np.arange(len(counts)))
        chunked = ex.reference_column_chunked(base, pos_lists)
        got_values, got_nulls = ex.export(chunked)
        flat = chunk_ids.astype(np.int64) * chunk + offsets
        want_null = nulls[flat] if with_nulls else np.zeros(len(flat), dtype=bool)
        np.testing.assert_array_equal(got_nulls.cpu().numpy().astype(bool), want_null)
        assert got_values.cpu().numpy().tobytes() == np.where(want_null, np_type(0), values[flat]).astype(np_type).tobytes()


@pytest.mark.parametrize("n", [0, 1, 4095, 100_001, 1_000_000])
def test_gather_row_ids(device, n):
    """hy_gather_row_ids: positions (chunk, offset) of a table of RowIDs presented in chunks -> the RowIDs there; NULL positions and
    positions past the table give the NULL RowID.  Small and odd counts (one RowID per thread) and large ones (four per thread)."""
    import torch
    from hyrise_amd.distributed import HipExecutor
    rng = np.random.default_rng(n)
    ex = HipExecutor(torch.device("cuda", 0))
    table_rows, chunk_rows = 70_001, 65_536
    table = rng.integers(0, 1 << 31, (table_rows, 2)).astype(np.int32)
    flat = rng.integers(0, table_rows + 500, n)                       # (some past the end)
    positions = np.stack([flat // chunk_rows, flat % chunk_rows], axis=1).astype(np.uint32)
    null_positions = rng.random(n) < 0.02
    positions[null_positions] = 0xFFFFFFFF
    got = ex.gather_row_ids(torch.from_numpy(table).to("cuda"), chunk_rows, torch.from_numpy(positions.view(np.int32)).to("cuda")).cpu().numpy()
    want = np.full((n, 2), -1, dtype=np.int32)
    ok = ~null_positions & (flat < table_rows)
    want[ok] = table[flat[ok]]
    np.testing.assert_array_equal(got.reshape(n, 2), want)
