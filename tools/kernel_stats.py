#!/usr/bin/env python3
"""Prints the per-kernel table of a rocprofv3 --kernel-trace --stats run (csv) found under a directory."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    raise SystemExit("no kernel_stats.csv under " + sys.argv[1])
for row in list(csv.DictReader(open(files[0])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(row["Name"][:70].ljust(70), row["Calls"].rjust(6), row["TotalDurationNs"].rjust(12), row["AverageNs"].rjust(12), row["Percentage"].rjust(7))
