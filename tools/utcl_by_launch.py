#!/usr/bin/env python3
"""Per pk_emit launch (dispatch order) of a rocprofv3 --pmc run: the counters' values.  usage: utcl_by_launch.py <dir>"""
import collections
import csv
import glob
import os
import sys

rows = collections.defaultdict(dict)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "pk_emit" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for i, dispatch in enumerate(sorted(rows)):
    print(i, dispatch, "  ".join(f"{k} {v:12.0f}" for k, v in sorted(rows[dispatch].items())))
