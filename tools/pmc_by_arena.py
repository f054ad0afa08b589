#!/usr/bin/env python3
"""Per-arena averages of a PMC run of tools/emit_lottery.py: python tools/pmc_by_arena.py <counter_collection.csv> <arenas> <launches per arena and measure() round>"""
import csv
import sys
from collections import defaultdict

path, arenas, per = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if "pk_emit" in r["Kernel_Name"]]
by_dispatch = defaultdict(dict)
for r in rows:
    by_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by_dispatch)
warm = 6 * arenas   # the joins before the first measure()
ids = ids[warm:]
for a in range(arenas):
    mine = ids[a * per:(a + 1) * per]
    names = sorted(by_dispatch[mine[0]]) if mine else []
    print(f"arena {a}: " + "  ".join(f"{n} {sum(by_dispatch[i][n] for i in mine) / len(mine):.0f}" for n in names))
