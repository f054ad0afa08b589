"""The C++ host mirror's LikeMatcher (two-pointer matching, hyrise_amd/host/hyrise_host.hpp) against the Python one
(anchored regex, hyrise_amd/like.py = the reference's sql_like_to_regex, like_matcher.cpp:32-55) on the reference's test
patterns and on random patterns over a small alphabet (many % and _, where a backtracking bug would show)."""
import os
import subprocess

import numpy as np
import pytest

from hyrise_amd import abi
from hyrise_amd.like import LikeMatcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def matcher_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("like") / "like_matcher"
    source = os.path.join(ROOT, "tests", "cpp", "like_matcher_main.cpp")
    library = os.path.join(ROOT, "hyrise_amd", "libhyrise_amd.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-o", str(out), source, library,
                           "-Wl,-rpath," + os.path.dirname(library)])
    return str(out)


def run(binary, cases):
    lines = "".join(f"{condition}\t{pattern.hex()}\t{text.hex()}\n" for condition, pattern, text in cases)
    result = subprocess.run([binary], input=lines.encode(), stdout=subprocess.PIPE, check=True)
    return [line == b"1" for line in result.stdout.split()]


def test_like_matcher_of_the_host_mirror(matcher_binary):
    rng = np.random.default_rng(5)
    alphabet = [b"a", b"b", b"A", b"%", b"_", b".", b"*", b"\\", b"(", b"\xc3\xa4"]
    cases = []
    for pattern in (b"%", b"Dampf%", b"%gesellschaft", b"%schifffahrtsgesellschaft%", b"Schiff%schaft", b"%D%_m_f%", b"D_m_f%", b"%2^2%", b"%$%$%",
                    b"%(%)%", b"%la\\.^$+?)({}.*__bl%", b"", b"_", b"%%", b"%_%_%"):
        for text in (b"", b"Dampfschifffahrtsgesellschaft", b"Schifffahrtsgesellschaft", b"Reeperbahn", b"1234", b"bla\\.^$+?)({}.*%_bla",
                     b"Something here!Money$$$2^2==_?4!something behind", b"Salamander(Cryptobranchoidea)", b"\xc3\xa4", b"ab"):
            for condition in (abi.PRED_LIKE, abi.PRED_NOT_LIKE, abi.PRED_LIKE_INSENSITIVE, abi.PRED_NOT_LIKE_INSENSITIVE):
                cases.append((condition, pattern, text))
    for _ in range(4000):
        pattern = b"".join(alphabet[i] for i in rng.integers(0, len(alphabet), rng.integers(0, 7)))
        text = b"".join(alphabet[i] for i in rng.choice([0, 1, 2, 5, 6, 7, 8, 9], rng.integers(0, 9)))
        cases.append((int(rng.choice([abi.PRED_LIKE, abi.PRED_NOT_LIKE, abi.PRED_LIKE_INSENSITIVE, abi.PRED_NOT_LIKE_INSENSITIVE])), pattern, text))
    got = run(matcher_binary, cases)
    want = [bool(LikeMatcher(pattern, condition)(text)) for condition, pattern, text in cases]
    wrong = [(c, g) for c, g, w in zip(cases, got, want) if g != w]
    assert not wrong, wrong[:5]
