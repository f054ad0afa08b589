#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by tools/profile_scan.sh into the two small files committed under profiles/:
scan_kernel_stats.csv (per-kernel call count / total / average duration) and scan_pmc.json (HBM bytes per scan_slices
launch).  FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B... see below; on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams, so it is doubled (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import sys


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def counter_per_launch(root, counter, kernel_substring):
    path = find(root, "counter_collection.csv")
    if not path:
        return None, 0
    total, launches = 0.0, 0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == counter and kernel_substring in row.get("Kernel_Name", ""):
                total += float(row["Counter_Value"])
                launches += 1
    return (total / launches if launches else None), launches


def main():
    out = sys.argv[1]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, "gpurun_out", "prof") if not os.path.isdir(out) else out
    stats = find(os.path.join(prof, "trace"), "kernel_stats.csv")
    summary = {}
    if stats:
        rows = list(csv.DictReader(open(stats)))
        keep = [r for r in rows if "hy::" in r.get("Name", "")]
        dst = os.path.join(prof, "scan_kernel_stats.csv")
        with open(dst, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(keep)
        for r in keep:
            if "scan_slices" in r["Name"]:
                summary["scan_slices_avg_ns"] = float(r.get("AverageNs", r.get("Average", 0)))
                summary["scan_slices_calls"] = int(float(r.get("Calls", 0)))
    fetch, n_f = counter_per_launch(os.path.join(prof, "pmc_fetch"), "FETCH_SIZE", "scan_slices")
    write, n_w = counter_per_launch(os.path.join(prof, "pmc_write"), "WRITE_SIZE", "scan_slices")
    summary.update({"FETCH_SIZE_per_launch_raw": fetch, "WRITE_SIZE_per_launch_raw": write, "launches_fetch": n_f, "launches_write": n_w})
    if fetch is not None and write is not None:
        # rocprofv3 reports both in kilobytes; FETCH_SIZE x2 on gfx950 for 16-byte-per-lane coalesced reads
        summary["fetch_bytes_corrected"] = fetch * 1024 * 2
        summary["write_bytes"] = write * 1024
        summary["hbm_bytes_per_launch"] = fetch * 1024 * 2 + write * 1024
    summary["note"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for wide coalesced streams); WRITE_SIZE uncalibrated; separate --pmc passes"
    with open(os.path.join(prof, "scan_pmc.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
