#!/usr/bin/env python3
"""Static look at one kernel's gfx950 assembly (tools/kernel_resources.py leaves the .s file): instruction mix of the whole kernel and of its
larger basic blocks -- where an instruction-bound loop spends its issue slots.  Usage: tools/isa_blocks.py FILE.s MANGLED_PREFIX [min block size]"""
import re
import sys
from collections import Counter


def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip().startswith(".Lfunc_end"))
    blocks, current = [], ["entry", Counter(), []]
    blocks.append(current)
    for raw in lines[start + 1:end]:
        l = raw.split(";")[0].strip()
        label = re.match(r"^(\.?[A-Za-z_][\w.$]*):$", l)
        if label:
            current = [label.group(1), Counter(), []]
            blocks.append(current)
            continue
        if not l or l.startswith("."):
            continue
        op = l.split()[0]
        current[1][klass(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            current[2].append(l.split()[-1])
    total = Counter()
    for _, mix, _ in blocks:
        total.update(mix)
    print("kernel:", dict(total), "instructions", sum(total.values()))
    for name, mix, targets in blocks:
        n = sum(mix.values())
        if n >= least:
            print(f"{name:14s} {n:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())) + ("   -> " + ",".join(targets) if targets else ""))


if __name__ == "__main__":
    main()
