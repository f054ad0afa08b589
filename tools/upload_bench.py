#!/usr/bin/env python3
"""hy_column_create from host segments: the three columns of the headline step (bench.timed_upload), twice each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
torch.cuda.init()
import bench  # noqa: E402
from hyrise_amd import abi, storage, tpch  # noqa: E402

lib = abi.load_library()
abi.check(lib.hy_init(0))
data = tpch.TpchData(10.0, 42)
for name, host in (("l_shipdate (Dictionary, u16 ids)", storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY)),
                   ("o_orderkey (ValueSegment<int32>)", storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)),
                   ("l_orderkey (FrameOfReference, u16)", storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))):
    for attempt in range(2):
        uploads = {}
        column = bench.timed_upload(name, host, uploads)
        entry = uploads[name]
        print(f"{name:38s} call {attempt}: {entry['ms']:8.2f} ms  {entry['achieved']:6.2f} GB/s  ({entry['bytes'] / 1e6:.0f} MB)", flush=True)
        column.close()
