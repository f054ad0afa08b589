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
