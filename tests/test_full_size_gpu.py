"""The BASELINE.json configurations at their FULL size (TPC-H SF10: 59 986 052 lineitem rows, 15 000 000 orders), checked
through properties that do not need the oracle to process 60 M rows: counts against numpy, order invariants of the
reference's output (PosLists ascending per chunk; join pairs grouped by radix partition and ascending in the probe
row inside a partition), key equality of EVERY join pair, and aggregate totals within the stated float tolerance."""
import ctypes as C

import numpy as np
import pytest

from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import aggregate_hash, make_predicate
from hyrise_amd.storage import DeviceColumn

pytestmark = pytest.mark.gpu


class DeviceArray:
    """A device buffer through the C ABI (hy_device_malloc / hy_memcpy_d2h): the tests need no torch."""

    def __init__(self, lib, shape, dtype):
        self.lib, self.shape, self.dtype = lib, shape, np.dtype(dtype)
        self.nbytes = int(np.prod(shape)) * self.dtype.itemsize
        pointer = C.c_void_p()
        abi.check(lib.hy_device_malloc(C.byref(pointer), max(self.nbytes, 256)))
        self.pointer = pointer.value

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        abi.check(self.lib.hy_memcpy_d2h(out.ctypes.data, self.pointer, self.nbytes))
        return out

    def __del__(self):
        if getattr(self, "pointer", None):
            self.lib.hy_device_free(self.pointer)
            self.pointer = None


@pytest.fixture(scope="module")
def sf10():
    return tpch.TpchData(scale_factor=10.0, seed=42)


def test_scan_sf10_shipdate(device, sf10):
    rows = sf10.n_lineitems
    host = storage.make_column(sf10.l_shipdate, None, abi.ENC_DICTIONARY)
    column = DeviceColumn(host)
    matches = DeviceArray(device, (rows, 2), np.uint32)
    offsets = DeviceArray(device, (host.n_chunks + 1,), np.int64)
    counts = DeviceArray(device, (host.n_chunks,), np.int32)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity = matches.pointer, rows
    result.offsets, result.counts = offsets.pointer, counts.pointer
    for condition, literal, literal2, expected in (
            (abi.PRED_LESS_THAN, tpch.DAY_1995_01_01, 0, sf10.l_shipdate < tpch.DAY_1995_01_01),
            (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, tpch.DAY_1994_01_01, tpch.DAY_1995_01_01, (sf10.l_shipdate >= tpch.DAY_1994_01_01) & (sf10.l_shipdate < tpch.DAY_1995_01_01))):
        predicate = make_predicate(condition, abi.TYPE_INT, literal, literal2)
        abi.check(device.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
        abi.check(device.hy_synchronize())
        got_counts = counts.numpy().astype(np.int64)
        chunk = abi.CHUNK_DEFAULT_SIZE
        per_chunk = np.add.reduceat(expected.astype(np.int64), np.arange(0, rows, chunk))
        np.testing.assert_array_equal(got_counts, per_chunk)                       # every chunk's count
        region = offsets.numpy()[:-1]
        np.testing.assert_array_equal(region, np.arange(host.n_chunks, dtype=np.int64) * chunk)
        # every RowID: concatenating the chunk regions gives exactly the matching row numbers, ascending
        host_matches = matches.numpy()
        keep = (np.arange(rows) % chunk) < np.repeat(got_counts, chunk)[:rows]      # the filled prefix of every region
        got_rows = host_matches[keep]
        np.testing.assert_array_equal(got_rows[:, 0].astype(np.int64) * chunk + got_rows[:, 1], np.flatnonzero(expected))


def test_join_sf10_orders_lineitem(device, sf10):
    orders = DeviceColumn(storage.make_column(sf10.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(sf10.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    n = sf10.n_lineitems
    left = DeviceArray(device, (n, 2), np.uint32)
    right = DeviceArray(device, (n, 2), np.uint32)
    slice_offsets = DeviceArray(device, (4096,), np.int64)
    r = abi.JoinResult()
    r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
    r.left_pos, r.right_pos, r.capacity = left.pointer, right.pointer, n
    r.slice_offsets, r.slice_capacity = slice_offsets.pointer, 4000
    abi.check(device.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
    abi.check(device.hy_synchronize())
    assert r.n_pairs == n and r.radix_bits == 7 and r.left_is_build == 1            # every lineitem has exactly one order
    chunk = abi.CHUNK_DEFAULT_SIZE
    build = left.numpy().astype(np.int64)
    probe = right.numpy().astype(np.int64)
    build_rows, probe_rows = build[:, 0] * chunk + build[:, 1], probe[:, 0] * chunk + probe[:, 1]
    np.testing.assert_array_equal(sf10.o_orderkey[build_rows], sf10.l_orderkey[probe_rows])        # EVERY pair joins equal keys
    partition = sf10.l_orderkey[probe_rows].astype(np.int64) & 127
    assert np.all(np.diff(partition) >= 0)                                          # pairs come radix partition by partition
    inside = np.diff(partition) == 0
    assert np.all(np.diff(probe_rows)[inside] > 0)                                  # ... ascending in the probe row inside one
    assert np.array_equal(np.sort(probe_rows), np.arange(n))                        # every probe row exactly once
    cuts = slice_offsets.numpy()[:r.n_slices + 1]
    assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts) > 0) and np.all(np.diff(cuts) <= 131070)
    per_partition = np.bincount(partition, minlength=128)
    assert r.n_slices == int(np.sum((per_partition + 131069) // 131070))           # a new PosList every 131 070 probe elements


def test_aggregate_sf10_q1_core(device, sf10):
    flag = DeviceColumn(storage.make_column(sf10.l_returnflag, None, abi.ENC_DICTIONARY))
    status = DeviceColumn(storage.make_column(sf10.l_linestatus, None, abi.ENC_DICTIONARY))
    quantity = DeviceColumn(storage.make_column(sf10.l_quantity, None, abi.ENC_UNENCODED))
    price = DeviceColumn(storage.make_column(sf10.l_extendedprice, None, abi.ENC_UNENCODED))
    got = aggregate_hash([flag, status], [(abi.AGG_SUM, quantity), (abi.AGG_AVG, price), (abi.AGG_COUNT, None), (abi.AGG_ANY, flag), (abi.AGG_ANY, status)],
                         group_capacity=64)
    keys = sf10.l_returnflag.astype(np.int64) * 256 + sf10.l_linestatus
    groups, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.argsort(first)                                                        # groups in order of first occurrence
    assert got.n_groups == len(groups)
    assert got.column(2) == counts[order].tolist()
    assert [f * 256 + s for f, s in zip(got.column(3), got.column(4))] == groups[order].tolist()
    for i, g in enumerate(groups[order]):
        members = keys == g
        total = float(sf10.l_quantity[members].astype(np.float64).sum())
        mean = float(sf10.l_extendedprice[members].astype(np.float64).mean())
        assert abs(got.column(0)[i] - total) <= 1e-9 * total and abs(got.column(1)[i] - mean) <= 1e-9 * mean
