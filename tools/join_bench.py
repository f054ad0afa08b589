import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hyrise_amd import abi, tpch, storage
from hyrise_amd.storage import DeviceColumn
lib = abi.load_library(); abi.check(lib.hy_init(0))
sf = float(os.environ.get("SF","10"))
t=time.time(); data = tpch.TpchData(scale_factor=sf, seed=42); print("gen", time.time()-t, data.n_orders, data.n_lineitems)
orders = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
lineitem = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
print("widths", lineitem.segments[0].width, "chunks", orders.n_chunks, lineitem.n_chunks)
do, dl = DeviceColumn(orders), DeviceColumn(lineitem)
dev = torch.device("cuda")
n = data.n_lineitems
left = torch.empty((n,2), dtype=torch.int32, device=dev); right = torch.empty((n,2), dtype=torch.int32, device=dev)
so = torch.zeros(2000, dtype=torch.int64, device=dev)
r = abi.JoinResult(); r.mem = abi.MEM_DEVICE; r.radix_bits = 0xFFFFFFFF; r.left_pos = left.data_ptr(); r.right_pos = right.data_ptr(); r.capacity = n; r.slice_offsets = so.data_ptr(); r.slice_capacity = 1990
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    abi.check(lib.hy_join_hash(do.handle, dl.handle, abi.JOIN_INNER, C.byref(r)))
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("join ms", dt*1e3, "pairs", r.n_pairs, "slices", r.n_slices, "radix", r.radix_bits, "rows/s %.3g" % ((data.n_orders+n)/dt))
