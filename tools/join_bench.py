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
r = abi.JoinResult(); r.mem = abi.MEM_DEVICE; r.radix_bits = int(os.environ.get('RADIX', '0xFFFFFFFF'), 0); r.left_pos = left.data_ptr(); r.right_pos = right.data_ptr(); r.capacity = n; r.slice_offsets = so.data_ptr(); r.slice_capacity = 1990
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    abi.check(lib.hy_join_hash(do.handle, dl.handle, abi.JOIN_INNER, C.byref(r)))
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("join ms", dt*1e3, "pairs", r.n_pairs, "slices", r.n_slices, "radix", r.radix_bits, "rows/s %.3g" % ((data.n_orders+n)/dt))
if os.environ.get("HY_JOIN_TRACE"):
    lib.hy_debug_join_trace.argtypes = [C.c_void_p, C.c_uint32]; lib.hy_debug_join_trace.restype = C.c_int
    buf = np.zeros((1 << 15, 6), dtype=np.uint64)
    nt = lib.hy_debug_join_trace(buf.ctypes.data, 1 << 15)
    t = buf[:nt].astype(np.int64)
    t = t[t[:, 5] > 0]
    d = np.diff(t, axis=1) / 100.0   # us
    names = ["evaluate", "clear+count", "prefix", "ranking+sync", "copy-out"] if os.environ.get("HY_JOIN_NO_RANK_TABLE") is None else ["loads+count", "prefix", "ranking", "sync", "copy-out"]
    print("probe_emit tiles", len(t), "kernel span us", (t[:, 5].max() - t[:, 0].min()) / 100.0)
    for i, nme in enumerate(names):
        print(f"  {nme:12s} mean {d[:, i].mean():7.2f} p50 {np.percentile(d[:, i], 50):7.2f} p90 {np.percentile(d[:, i], 90):7.2f}")
    print("  total        mean %.2f" % ((t[:, 5] - t[:, 0]).mean() / 100.0))
    per = (nt + 7) // 8
    for x in range(8):   # tiles are handed out per XCD (tickets): when does each XCD finish its share?
        sel = t[(np.arange(nt)[buf[:nt, 5] > 0] // per) == x]
        print("  xcd %d: %5d tiles, first start %.1f last end %.1f us, tile mean %.2f" % (x, len(sel), (sel[:, 0].min() - t[:, 0].min()) / 100.0,
              (sel[:, 5].max() - t[:, 0].min()) / 100.0, (sel[:, 5] - sel[:, 0]).mean() / 100.0))
