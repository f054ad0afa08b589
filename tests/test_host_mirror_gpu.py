"""Runs the C++ mirror of Hyrise's operator interface (hyrise_amd/host/hyrise_host.hpp) through the reference's own
operator tests, re-stated in tests/cpp/host_tests.cpp.  The binary links libhyrise_amd.so; this Python process loads
the same library first so that a missing build fails here, loudly."""
import os
import subprocess

import pytest

from hyrise_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_operator_interface(device):
    binary = os.path.join(ROOT, "tests", "cpp", "host_tests")
    assert os.path.exists(binary), "tests/cpp/host_tests missing: run __graft_entry__.build()"
    proc = subprocess.run([binary, os.path.join(ROOT, "tests", "golden", "tbl")], capture_output=True, text=True, timeout=300)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "HOST TESTS PASSED" in proc.stdout


def test_cpp_operator_chain_sf10_stays_in_hbm(device):
    """TableScan -> JoinHash -> AggregateHash through `_on_execute()` at TPC-H SF10 size with DevicePosLists between the operators
    (tests/cpp/operator_chain.cpp): every PosList of every output table byte-equal to the host-result form of the same chain and to the
    oracle's operators (hyo_table_scan / hyo_join_hash / hyo_aggregate_hash, loaded by the binary with dlopen)."""
    binary = os.path.join(ROOT, "tests", "cpp", "operator_chain")
    oracle = os.path.join(ROOT, "oracle", "liboracle.so")
    assert os.path.exists(binary), "tests/cpp/operator_chain missing: run __graft_entry__.build()"
    assert os.path.exists(oracle), "oracle/liboracle.so missing: run __graft_entry__.build()"
    threads = str(max(1, min(32, os.cpu_count() or 1)))
    proc = subprocess.run([binary, "--oracle", oracle, "--threads", threads], capture_output=True, text=True, timeout=1500)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "OPERATOR CHAIN OK" in proc.stdout
    for line in ("TableScan vs oracle", "JoinHash vs oracle", "AggregateHash vs oracle", "JoinHash output, device-resident vs host-result"):
        assert line in proc.stdout, line
