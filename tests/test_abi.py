"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol the header declares."""
import os
import re

import pytest

from hyrise_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "hyrise_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hy_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(abi.LIB_PATH), "libhyrise_amd.so missing: run __graft_entry__.build()"


def test_every_declared_symbol_is_exported_and_bound():
    lib = abi.load_library()
    declared = header_functions()
    bound = sorted(name for name, _, _ in abi.SYMBOLS)
    assert declared == bound, f"header and abi.py disagree: {set(declared) ^ set(bound)}"
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"


def test_abi_version_and_struct_sizes():
    lib = abi.load_library()
    assert lib.hy_abi_version() == 4
    import ctypes as C
    assert C.sizeof(abi.RowID) == 8          # types.hpp:97-117
    assert C.sizeof(abi.Segment) == 64
    assert C.sizeof(abi.Value) == 8


def test_calls_without_a_device_fail_loudly():
    """No GPU in the build container: the product path must say so instead of computing anything on the CPU."""
    import ctypes as C
    lib = abi.load_library()
    count = C.c_int32(-1)
    lib.hy_device_count(C.byref(count))
    if count.value > 0:
        pytest.skip("a GPU is visible")
    status = lib.hy_init(0)
    assert status == abi.ERR_DEVICE
    assert lib.hy_last_error()


def test_header_is_valid_c_and_cxx_and_ctypes_layouts_match(tmp_path):
    """The boundary is a C ABI: include/hyrise_amd.h must compile as C11 and as C++17 on its own, and the ctypes mirror of
    every struct (hyrise_amd/abi.py) must have the size the C compiler gives it."""
    import ctypes as C
    import subprocess
    header = os.path.join(ROOT, "include", "hyrise_amd.h")
    for compiler, flags in (("gcc", ["-std=c11", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        subprocess.check_call([compiler] + flags + ["-Wall", "-Wextra", "-Werror", "-fsyntax-only", header])
    structs = {"hy_row_id": abi.RowID, "hy_segment": abi.Segment, "hy_value": abi.Value, "hy_predicate": abi.Predicate,
               "hy_scan_result": abi.ScanResult, "hy_join_predicate": abi.JoinPredicate, "hy_join_result": abi.JoinResult, "hy_join_status": abi.JoinStatus, "hy_lz4_blocks": abi.Lz4Blocks, "hy_operand": abi.Operand,
               "hy_aggregate_spec": abi.AggregateSpec, "hy_aggregate_column": abi.AggregateColumn, "hy_aggregate_result": abi.AggregateResult,
               "hy_expression_node": abi.ExpressionNode, "hy_expression": abi.Expression, "hy_filter": abi.Filter, "hy_fused_aggregate": abi.FusedAggregate,
               "hy_star_dimension": abi.StarDimension, "hy_star_column": abi.StarColumn, "hy_star_aggregate": abi.StarAggregate}
    source = tmp_path / "sizes.c"
    lines = [f'  printf("{name} %zu\\n", sizeof({name}));\n' for name in structs]
    for name, mirror in structs.items():   # ... and every field sits where the C compiler puts it (same field names on both sides)
        lines += [f'  printf("{name}.{field[0]} %zu\\n", offsetof({name}, {field[0]}));\n' for field in mirror._fields_]
    source.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "hyrise_amd.h"\nint main(void) {\n' + "".join(lines) + "  return 0;\n}\n")
    binary = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(binary), str(source)])
    sizes = dict(line.split() for line in subprocess.run([str(binary)], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines())
    for name, mirror in structs.items():
        assert int(sizes[name]) == C.sizeof(mirror), f"{name}: C says {sizes[name]}, ctypes {C.sizeof(mirror)}"
        for field in mirror._fields_:
            assert int(sizes[f"{name}.{field[0]}"]) == getattr(mirror, field[0]).offset, f"{name}.{field[0]}"
