#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs of tools/collect_profiles.sh into the files committed under profiles/:
  r02_<leg>_kernel_stats.csv   the hy:: rows of rocprofv3's kernel_stats.csv (calls, total, average, share)
  r02_<leg>_pmc.json           per kernel: launches and FETCH_SIZE / WRITE_SIZE per launch (KB as rocprofv3 reports them), and
                               `hbm_bytes_per_launch` as bench.py's roofline.traffic reads it: for scan the bytes of one
                               scan_slices launch; for join / aggregate the bytes of ALL kernels of one operator call.
FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes, MI355X_MICROARCH.md "HBM");
WRITE_SIZE is taken as reported (uncalibrated in the guide).  Both include Infinity-Cache hits."""
import collections
import csv
import glob
import json
import os
import sys

DOMINANT = {"scan": "scan_slices", "join": "rt_probe_emit", "aggregate": "aggregate_rows", "fused": "fused_rows"}


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("hy::", "").replace("(anonymous namespace)::", "").strip()


def counters(root, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and "hy::" in row.get("Kernel_Name", ""):
                    cell = per[short(row["Kernel_Name"])]
                    cell[0] += float(row["Counter_Value"])
                    cell[1] += 1
    return per


def main():
    out, legs = sys.argv[1], sys.argv[2:]
    for leg in legs:
        stats = glob.glob(os.path.join(out, leg, "trace", "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            rows = list(csv.DictReader(open(stats[0])))
            keep = [r for r in rows if "hy::" in r.get("Name", "")]
            with open(os.path.join(out, f"r02_{leg}_kernel_stats.csv"), "w", newline="") as fh:
                writer = csv.DictWriter(fh, fieldnames=rows[0].keys())
                writer.writeheader()
                writer.writerows(keep)
            for r in keep[:12]:
                print(f"{leg:10s} {short(r['Name'])[:48]:48s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']:>6s}%")
        fetch, write = counters(os.path.join(out, leg, "fetch"), "FETCH_SIZE"), counters(os.path.join(out, leg, "write"), "WRITE_SIZE")
        kernels = {}
        for name in sorted(set(fetch) | set(write)):
            f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
            kernels[name] = {"launches": max(f[1], w[1]), "FETCH_SIZE_KB_per_launch": f[0] / f[1] if f[1] else None,
                             "WRITE_SIZE_KB_per_launch": w[0] / w[1] if w[1] else None,
                             "hbm_bytes_per_launch": (f[0] / f[1] * 2048 if f[1] else 0) + (w[0] / w[1] * 1024 if w[1] else 0)}
        if not kernels:   # (a leg collected without counter passes)
            continue
        summary = {"leg": leg, "dominant_kernel": DOMINANT[leg], "kernels": kernels,
                   "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); bytes = 2 x FETCH_SIZE KB x 1024 + WRITE_SIZE KB x 1024; Infinity-Cache hits included"}
        dominant = [k for k in kernels if DOMINANT[leg] in k]
        if leg == "scan":
            summary["hbm_bytes_per_launch"] = kernels[dominant[0]]["hbm_bytes_per_launch"] if dominant else None
        else:
            # one operator call = the dominant kernel's launch count; every other kernel's bytes are spread over those calls
            calls = max((kernels[k]["launches"] for k in dominant), default=0)
            total = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in kernels.values())
            summary["operator_calls"] = calls
            summary["hbm_bytes_per_launch"] = total / calls if calls else None
            summary["dominant_kernel_hbm_bytes_per_launch"] = sum(kernels[k]["hbm_bytes_per_launch"] for k in dominant) if dominant else None
        with open(os.path.join(out, f"r02_{leg}_pmc.json"), "w") as fh:
            json.dump(summary, fh, indent=1)
        print(leg, "hbm_bytes_per_launch", summary["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
