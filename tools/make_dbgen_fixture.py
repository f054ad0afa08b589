#!/usr/bin/env python3
"""Writes tests/golden/dbgen/tpch_sf0.02.npz: orders / lineitem rows of the reference's vendored dbgen at scale factor 0.02 (30 000
orders, 120 515 lineitems = two chunks), generated HERE by oracle/_ref/tpch_rows (`make -C oracle ref`: third_party/tpch-dbgen compiled
from the reference tree, driven like TPCHTableGenerator::generate) and stored in narrow integer types (dates as days since 1992-01-01,
money as cents).  hyrise_amd.tpch.DbgenData.from_fixture() turns them into Hyrise's column types.  Run from the repo root."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyrise_amd.tpch import DbgenData  # noqa: E402

SCALE_FACTOR = "0.02"


def main():
    binary = os.path.join(ROOT, "oracle", "_ref", "tpch_rows")
    if not os.path.exists(binary):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "rows.bin")
        subprocess.check_call([binary, SCALE_FACTOR, path])
        arrays = DbgenData.read_rows_file(path)
    narrow = {"o_orderkey": np.int32, "l_orderkey": np.int32, "l_quantity": np.uint8, "l_extendedprice_cents": np.int32, "l_discount_cents": np.uint8,
              "l_tax_cents": np.uint8, "l_returnflag": np.uint8, "l_linestatus": np.uint8, "l_shipdate": np.uint16, "l_commitdate": np.uint16, "l_receiptdate": np.uint16}
    out = {}
    for name, dtype in narrow.items():
        out[name] = arrays[name].astype(dtype)
        assert np.array_equal(out[name].astype(np.int64), arrays[name].astype(np.int64)), name
    target = os.path.join(ROOT, "tests", "golden", "dbgen")
    os.makedirs(target, exist_ok=True)
    np.savez_compressed(os.path.join(target, f"tpch_sf{SCALE_FACTOR}.npz"), **out)
    print(f"{len(out['o_orderkey'])} orders, {len(out['l_orderkey'])} lineitems -> {target}")


if __name__ == "__main__":
    main()
