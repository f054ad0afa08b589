#!/usr/bin/env python3
"""Copies the reference's own test fixtures for the hot path into tests/golden/ (they are test DATA, not source), and
records provenance.  Run in the build container where /root/reference is mounted:  python tests/golden/make_golden.py

The expected values that the reference keeps inline in its gtest sources (explicit multisets in table_scan_test.cpp,
index lists in table_scan_between_test.cpp, Bloom/histogram known answers in join_hash_steps_test.cpp) are restated
with file:line citations in tests/golden/known_answers.py.
"""
import hashlib
import json
import os
import shutil

REF = "/root/reference/resources/test_data/tbl"
HERE = os.path.dirname(os.path.abspath(__file__))

FILES = [
    # TableScan fixtures (src/test/lib/operators/table_scan_test.cpp, table_scan_between_test.cpp)
    "int_float.tbl", "int_float2.tbl", "int_float_filtered.tbl", "int_float_filtered2.tbl", "int_float_with_null.tbl",
    "int_int_shuffled.tbl", "int_int_shuffled_2.tbl", "int_sorted.tbl", "int_int_w_null_8_rows.tbl",
    "int_int3.tbl", "int_int_int.tbl", "int_string_like.tbl", "int_float_null_1.tbl", "int_float_null_2.tbl",
    "int_float4.tbl", "int_float_double_string.tbl", "float_int.tbl", "int.tbl", "int2.tbl", "int3.tbl",
    # LIKE on dictionary segments: inputs and expected outputs of src/test/lib/operators/table_scan_string_test.cpp:66-327
    "int_string_like_starting.tbl", "int_string_like_ending.tbl", "int_string_like_containing.tbl",
    "int_string_like_containing_wildcard.tbl", "int_string_like_without_null.tbl", "int_string_like_not_starting.tbl",
    "int_string_like_equals.tbl", "int_string_like_not_equals.tbl", "int_string_like_less_than.tbl",
    "int_string_like_special_chars.tbl", "int_string_like_special_chars_1.tbl", "int_string_like_special_chars_2.tbl",
    "int_string_like_special_chars_3.tbl",
    # JoinHash: differential-testing inputs (src/test/lib/operators/join_test_runner.cpp:656-791)
    "join_test_runner/input_table_left_0.tbl", "join_test_runner/input_table_left_10.tbl",
    "join_test_runner/input_table_left_15.tbl", "join_test_runner/input_table_right_0.tbl",
    "join_test_runner/input_table_right_10.tbl", "join_test_runner/input_table_right_15.tbl",
    # realistic inputs (join_hash_test.cpp:24-31)
    "tpch/sf-0.001/lineitem.tbl", "tpch/sf-0.001/orders.tbl",
    # Projection arithmetic: the ExpressionEvaluator's series tests (src/test/lib/expression/expression_evaluator_to_values_test.cpp:38,244-256)
    "expression_evaluator/input_a.tbl",
    # Projection: int_float.tbl's a + b (src/test/lib/operators/projection_test.cpp:59-64)
    "projection/int_float_add.tbl",
]


# Binary tables written by Hyrise itself (src/test/lib/import_export/binary/binary_writer_test.cpp compares its output
# byte by byte with these files): real DictionarySegment / FrameOfReferenceSegment / ValueSegment layouts.
BIN_REF = "/root/reference/resources/test_data/bin"
BIN_FILES = ["SingleChunkFrameOfReferenceSegment.bin", "MultipleChunksFrameOfReferenceSegment.bin", "NullValuesFrameOfReferenceSegment.bin",
             "AllNullFrameOfReferenceSegment.bin", "SortColumnDefinitions.bin", "TwoColumnsNoValues.bin", "float.bin", "int_float.bin",
             "int_float_deleted.bin", "int_string2.bin", "FixedStringDictionarySingleChunk.bin", "FixedStringDictionaryNullValue.bin",
             "FixedStringDictionaryMultipleChunks.bin", "LZ4MultipleBlocks.bin"]
BIN_DIRS = ["AllTypesAllNullValues", "AllTypesMixColumn", "AllTypesNullValues", "AllTypesSegmentSorted", "AllTypesSegmentUnsorted",
            "EmptyStringsSegment", "MultipleChunkSingleFloatColumn", "RepeatedInt", "RunNullValues", "SingleChunkSingleFloatColumn", "StringSegment"]


def copy_binary_tables(manifest):
    names = list(BIN_FILES)
    for directory in BIN_DIRS:
        names += [f"{directory}/{encoding}.bin" for encoding in ("Unencoded", "Dictionary", "RunLength", "LZ4")]
    for name in names:
        src = os.path.join(BIN_REF, name)
        if not os.path.exists(src):
            print("missing in reference:", name)
            continue
        dst = os.path.join(HERE, "bin", name)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        with open(src, "rb") as fh:
            manifest["bin/" + name] = {"source": "resources/test_data/bin/" + name, "sha256": hashlib.sha256(fh.read()).hexdigest()}


def main():
    manifest = {}
    copy_binary_tables(manifest)
    names = list(FILES)
    # AggregateHash: every input/expected pair (src/test/lib/operators/aggregate_test.cpp:290-852)
    for root, _, files in os.walk(os.path.join(REF, "aggregateoperator")):
        for f in sorted(files):
            names.append(os.path.relpath(os.path.join(root, f), REF))
    for name in names:
        src = os.path.join(REF, name)
        if not os.path.exists(src):
            print("missing in reference:", name)
            continue
        dst = os.path.join(HERE, "tbl", name)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        with open(src, "rb") as fh:
            manifest[name] = {"source": "resources/test_data/tbl/" + name, "sha256": hashlib.sha256(fh.read()).hexdigest()}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print(len(manifest), "fixtures copied")


if __name__ == "__main__":
    main()


def extract_aggregate_cases():
    """Restates the (input table, aggregates, GROUP BY columns, expected .tbl) tuples of every `test_output<TypeParam>`
    call in src/test/lib/operators/aggregate_test.cpp (with the line it comes from) into aggregate_cases.json."""
    import re
    src_path = "/root/reference/src/test/lib/operators/aggregate_test.cpp"
    text = open(src_path).read()
    wrappers = {}
    for m in re.finditer(r"(_table_wrapper_\w+) = std::make_shared<TableWrapper>\(\s*load_table\(\"resources/test_data/tbl/([^\"]+)\", ChunkOffset\{(\d+)\}\)\)", text):
        wrappers[m.group(1)] = {"input": m.group(2), "chunk_size": int(m.group(3)), "encoded": False}
    # dictionary-encoded variants are built from a local `test_table`
    wrappers["_table_wrapper_1_1_dict"] = {"input": "aggregateoperator/groupby_int_1gb_1agg/input.tbl", "chunk_size": 2, "encoded": True}
    wrappers["_table_wrapper_1_1_null_dict"] = {"input": "aggregateoperator/groupby_int_1gb_1agg/input_null.tbl", "chunk_size": 2, "encoded": True}
    cases = []
    for m in re.finditer(r"test_output<TypeParam>\(", text):
        end = text.index(");", m.end())
        call = text[m.end():end]
        # split top-level arguments
        args, depth, cur = [], 0, ""
        for ch in call:
            if ch in "{(":
                depth += 1
            elif ch in "})":
                depth -= 1
            if ch == "," and depth == 0:
                args.append(cur.strip())
                cur = ""
            else:
                cur += ch
        args.append(cur.strip())
        wrapper = args[0].replace("this->", "")
        if wrapper not in wrappers or len(args) < 4:
            continue
        aggregates = [(int(a), f) for a, f in re.findall(r"\{ColumnID\{(\d+)\}, WindowFunction::(\w+)\}", args[1])]
        aggregates += [(None, f) for f in re.findall(r"\{INVALID_COLUMN_ID, WindowFunction::(\w+)\}", args[1])]
        groupby = [int(g) for g in re.findall(r"ColumnID\{(\d+)\}", args[2])]
        expected = re.search(r"resources/test_data/tbl/([^\"]+)", args[3]).group(1)
        line = text[:m.start()].count("\n") + 1
        cases.append({"line": line, **wrappers[wrapper], "aggregates": aggregates, "groupby": groupby, "expected": expected})
    with open(os.path.join(HERE, "aggregate_cases.json"), "w") as fh:
        json.dump({"source": "src/test/lib/operators/aggregate_test.cpp", "cases": cases}, fh, indent=1)
    print(len(cases), "aggregate cases restated")


if __name__ == "__main__":
    extract_aggregate_cases()
