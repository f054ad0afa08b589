"""The C++ coordinator of the sharded operators (hyrise_amd/host/multi_gpu.hpp) through real RCCL calls: one process, one worker thread
and one communicator per device (hy_comm_init_all = ncclCommInitAll).  tests/cpp/multi_gpu_tests.cpp compares every sharded result with
the single-GPU operator and with a nested loop.  On a one-GPU box the world is 1 (every collective still goes through librccl)."""
import os
import subprocess

import pytest

from hyrise_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "tests", "cpp", "multi_gpu_tests")


def test_cpp_coordinator_on_every_device_of_the_box(device):
    assert os.path.exists(BINARY), "tests/cpp/multi_gpu_tests missing: run __graft_entry__.build()"
    proc = subprocess.run([BINARY], capture_output=True, text=True, timeout=600)
    print(proc.stdout)
    print(proc.stderr[-3000:])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "MULTI GPU TESTS PASSED" in proc.stdout


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_that_share_one_device(device, ranks):
    """ncclCommInitAll refuses a device list that names one GPU twice; worker threads that share a GPU exchange through its memory
    (csrc/comm.hip LocalExchange) behind the same hy_comm_* entry points.  The coordinator's world > 1 logic -- count matrices,
    rank-ordered merges, chunk offsets -- runs here on a one-GPU box; the RCCL transport itself is exercised at world 1 above and at
    world N on a multi-GPU box."""
    proc = subprocess.run([BINARY] + ["0"] * ranks, capture_output=True, text=True, timeout=600)
    print(proc.stdout)
    print(proc.stderr[-3000:])
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert f"(world {ranks})" in proc.stdout and "MULTI GPU TESTS PASSED" in proc.stdout
