"""The C++ host mirror's load_table + ChunkEncoder (hyrise_amd/host/hyrise_host.hpp) against the Python encoders
(hyrise_amd/storage.py, which the Hyrise .bin exports under tests/golden/bin pin): same widths, same attribute vectors /
offsets / minima / dictionaries / null words, byte for byte, on every .tbl fixture at several chunk sizes.  CPU only."""
import glob
import os
import subprocess

import numpy as np
import pytest

from hyrise_amd import abi, storage
from support import load_tbl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "tbl", "*.tbl")) +
                glob.glob(os.path.join(ROOT, "tests", "golden", "tbl", "aggregateoperator", "*", "input.tbl")))
ENCODINGS = {"unencoded": abi.ENC_UNENCODED, "dictionary": abi.ENC_DICTIONARY, "for": abi.ENC_FRAME_OF_REFERENCE}


@pytest.fixture(scope="module")
def dump_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("encoders") / "encode_dump"
    source = os.path.join(ROOT, "tests", "cpp", "encode_dump_main.cpp")
    library = os.path.join(ROOT, "hyrise_amd", "libhyrise_amd.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-o", str(out), source, library,
                           "-Wl,-rpath," + os.path.dirname(library)])
    return str(out)


def hex_of(array):
    return "-" if array is None or len(array) == 0 else np.ascontiguousarray(array).tobytes().hex()


def string_entries(entries):
    return "-" if len(entries) == 0 else ",".join(e.hex() if e else "." for e in entries)


def expected_lines(table, chunk_size, encoding):
    lines = []
    for begin in range(0, table.rows, chunk_size):
        end = min(table.rows, begin + chunk_size)
        chunk_id = begin // chunk_size
        for c, data_type in enumerate(table.types):
            values = table.columns[c][begin:end]
            nulls = table.nulls[c][begin:end] if table.nullable[c] else None
            enc = encoding
            if enc == abi.ENC_FRAME_OF_REFERENCE and data_type != abi.TYPE_INT:
                enc = abi.ENC_UNENCODED
            if data_type == abi.TYPE_STRING:
                raw = [v.encode() for v in values]
                if enc == abi.ENC_DICTIONARY:
                    segment, dictionary = storage.encode_string_dictionary(raw, nulls)
                    fields = ["dictionary", segment.size, segment.width, hex_of(segment.data), string_entries(dictionary), "-"]
                else:
                    words = storage.pack_nulls(nulls) if nulls is not None else None
                    fields = ["value", len(raw), 32, string_entries(raw), "-", hex_of(words)]
            else:
                segment = storage.encode_segment(values, nulls, enc)
                kind = {abi.ENC_UNENCODED: "value", abi.ENC_DICTIONARY: "dictionary", abi.ENC_FRAME_OF_REFERENCE: "for"}[enc]
                fields = [kind, segment.size, segment.width, hex_of(segment.data), hex_of(segment.aux), hex_of(segment.nulls)]
            lines.append(" ".join(str(f) for f in [chunk_id, c] + fields))
    return lines


@pytest.mark.parametrize("encoding", sorted(ENCODINGS))
def test_host_mirror_encodes_like_the_python_encoders(dump_binary, encoding):
    assert len(TABLES) > 25
    checked = 0
    for path in TABLES:
        table = load_tbl(path)
        for chunk_size in (1, 2, 3, 7, 100000):
            got = subprocess.run([dump_binary, path, str(chunk_size), encoding], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
            want = expected_lines(table, chunk_size, ENCODINGS[encoding])
            assert got == want, (path, chunk_size, [(g, w) for g, w in zip(got, want) if g != w][:2], len(got), len(want))
            checked += len(got)
    assert checked > 1000


def test_host_mirror_encoders_on_a_generated_table_with_wide_vectors(dump_binary, tmp_path):
    """70 000 rows: u16 / u32 attribute vectors and offsets, several FrameOfReference blocks, NULLs at word boundaries."""
    rng = np.random.default_rng(11)
    n = 70000
    columns = {
        "a|int_null": rng.integers(-2**31, 2**31, n),            # FoR offsets need u32
        "b|int": rng.integers(1000, 1000 + 40000, n),            # dictionary ids need u16, FoR offsets u16
        "c|long_null": rng.integers(-2**62, 2**62, n),
        "d|double": np.round(rng.normal(size=n), 3) + 0.0,   # + 0.0: no -0.0 (which of +-0.0 a dictionary keeps is unspecified)
        "e|float_null": rng.integers(0, 300, n).astype(np.float32) / 4,
        "f|string_null": rng.integers(0, 5000, n),
    }
    nulls = {name: rng.random(n) < 0.1 for name in columns if name.endswith("_null")}
    for mask in nulls.values():
        mask[[0, 63, 64, 127, 128, n - 1]] = True
    path = tmp_path / "generated.tbl"
    with open(path, "w") as fh:
        fh.write("|".join(name.split("|")[0] for name in columns) + "\n")
        fh.write("|".join(name.split("|")[1] for name in columns) + "\n")
        cells = []
        for name, values in columns.items():
            if name.startswith("f"):
                text = np.array([f"key#{v:05d}" for v in values], dtype=object)
            elif values.dtype.kind == "f":
                text = np.array([repr(float(v)) for v in values], dtype=object)
            else:
                text = values.astype(str).astype(object)
            if name in nulls:
                text[nulls[name]] = "null"
            cells.append(text)
        for row in zip(*cells):
            fh.write("|".join(row) + "\n")
    table = load_tbl(str(path))
    for encoding, chunk_size in (("dictionary", 65535), ("for", 65535), ("unencoded", 65535), ("dictionary", 4100), ("for", 2048)):
        got = subprocess.run([dump_binary, str(path), str(chunk_size), encoding], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        want = expected_lines(table, chunk_size, ENCODINGS[encoding])
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g == w, (encoding, chunk_size, g[:80], w[:80])
        widths = {line.split(" ")[4] for line in got if line.split(" ")[2] == "for"}
        if encoding == "for" and chunk_size == 65535:
            assert widths == {"2", "4"}, widths
