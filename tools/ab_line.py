#!/usr/bin/env python3
"""One bench.py line (stdin) -> step time and the step's kernel times, for tools/r5_ab.sh."""
import json
import sys

d = json.loads(sys.stdin.read())
kernels = d["roofline"]["kernels"]
print("step_ms %.4f " % d["ms_per_step"] + " ".join("%s=%.1fus" % (name, v["kernel_ms"] * 1e3) for name, v in kernels.items()))
