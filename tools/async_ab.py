#!/usr/bin/env python3
"""The headline step (TableScan + JoinHash, SF10) with and without host round trips, in ONE process: does a GPU that never idles run its
kernels as fast as one that rests between joins?  Prints ms per step and the HIP-event times of the step's kernels for every mode, and
samples the device's clock and power (sysfs) while a long asynchronous and a long synchronous loop run.
Usage: python tools/async_ab.py [steps]"""
import ctypes as C
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sysfs_sampler(stop, out):
    clocks = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    powers = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
    while not stop.is_set():
        sample = []
        for path in clocks[:1]:
            try:
                active = [line for line in open(path).read().splitlines() if line.endswith("*")]
                sample.append(active[0] if active else "?")
            except OSError:
                pass
        for path in powers[:1]:
            try:
                sample.append(f"{int(open(path).read()) / 1e6:.0f} W")
            except (OSError, ValueError):
                pass
        out.append(sample)
        time.sleep(0.05)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.operators import make_predicate
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    abi.check(lib.hy_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    dev = torch.device("cuda", 0)
    rows = tpch.LINEITEM_ROWS_SF10
    days, host_column = tpch.shipdate_column(rows, seed=42)
    columns = [DeviceColumn(host_column) for _ in range(3)]
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders_host = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem_host = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    orders = [DeviceColumn(orders_host) for _ in range(3)]
    lineitem = [DeviceColumn(lineitem_host) for _ in range(3)]
    n = data.n_lineitems
    matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    offsets = torch.zeros(host_column.n_chunks + 1, dtype=torch.int64, device=dev)
    counts = torch.zeros(host_column.n_chunks, dtype=torch.int32, device=dev)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity, result.offsets, result.counts = matches.data_ptr(), rows, offsets.data_ptr(), counts.data_ptr()
    turn = [0]

    def scan():
        abi.check(lib.hy_table_scan(columns[turn[0] % 3].handle, C.byref(predicate), None, 0, C.byref(result)))
        turn[0] += 1

    join_async, r_async, keep_a = bench.device_join(lib, torch, dev, orders, lineitem, n, asynchronous=True)
    join_sync, r_sync, keep_s = bench.device_join(lib, torch, dev, orders, lineitem, n, asynchronous=False)
    for _ in range(6):
        join_sync()

    def loop(name, body, k, every=4):
        for _ in range(3):
            body()
        abi.check(lib.hy_set_profiling(every))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            body()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        kinds = bench.kernel_times(lib)
        abi.check(lib.hy_set_profiling(0))
        print(f"{name:46s} {dt * 1e3:7.4f} ms/step  " + "  ".join(f"{key} {v[0] * 1e3:6.1f}" for key, v in kinds.items() if v[1]), flush=True)
        return dt

    def both_async():
        scan(); join_async()

    def both_sync():
        scan(); join_sync()

    def async_then_wait():
        scan(); join_async(); lib.hy_synchronize()

    pace = [0]

    def async_wait_every_2():
        scan(); join_async()
        pace[0] += 1
        if pace[0] % 2 == 0:
            lib.hy_synchronize()

    for round_ in range(2):
        loop("scan + join, async", both_async, steps)
        loop("scan + join, sync", both_sync, steps)
        loop("scan + join, async + hy_synchronize per step", async_then_wait, steps)
        loop("scan + join, async + hy_synchronize per 2 steps", async_wait_every_2, steps)
        loop("join only, async", join_async, steps)
        loop("join only, sync", join_sync, steps)
        loop("scan only", scan, steps)
        loop("scan + join, async, no event pairs", both_async, steps, every=0)
        loop("scan + join, sync, no event pairs", both_sync, steps, every=0)
    for name, body in (("async", both_async), ("sync", both_sync), ("async", both_async)):
        stop, samples = threading.Event(), []
        sampler = threading.Thread(target=sysfs_sampler, args=(stop, samples))
        sampler.start()
        dt = loop(f"long loop, {name}", body, 3000, every=0)
        stop.set()
        sampler.join()
        print(f"  {name}: sysfs samples (clock | power) {samples[2:-1][:12]} ... {len(samples)} samples", flush=True)
    join_async.finish()


if __name__ == "__main__":
    main()
