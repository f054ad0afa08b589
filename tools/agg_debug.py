import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from hyrise_amd import abi, tpch
from hyrise_amd.operators import aggregate_hash
from hyrise_amd.storage import DeviceColumn
lib = abi.load_library(); abi.check(lib.hy_init(0))
data = tpch.TpchData(10.0, 42)
g, m, _ = tpch.q1_core_columns(data)
gd = [DeviceColumn(c) for c in g]; md = {k: DeviceColumn(v) for k, v in m.items()}
spec = [(abi.AGG_SUM, md["l_quantity"]), (abi.AGG_SUM, md["l_extendedprice"]), (abi.AGG_AVG, md["l_quantity"]), (abi.AGG_AVG, md["l_extendedprice"]), (abi.AGG_AVG, md["l_discount"]), (abi.AGG_COUNT, None)]
for name, env in (("default", {}), ("two histograms instead of the pair histogram", {"HY_AGG_NO_JOINT_HISTOGRAM": "1"}), ("default again", {}), ("no histograms", {"HY_AGG_SMALL_DEBUG": "1"}), ("no 2-byte column", {"HY_AGG_SMALL_DEBUG": "2"}),
                  ("no dense lookup", {"HY_AGG_SMALL_DEBUG": "8"}), ("no histograms, no 2-byte column, no dense lookup", {"HY_AGG_SMALL_DEBUG": "11"}), ("generic kernel", {"HY_AGG_NO_SMALL_DOMAIN": "1"})):
    _switches = abi.switches(env)
    _switches.__enter__()
    dt, km = bench.timed_kernel(lib, torch, lambda: aggregate_hash(gd, spec, group_capacity=64), 5, kind="aggregate")
    print(f"{name:24s} {dt*1e3:7.3f} ms/call  kernel {km*1e3:7.1f} us", flush=True)
    _switches.__exit__(None, None, None)
