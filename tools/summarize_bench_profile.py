#!/usr/bin/env python3
"""Summarises the rocprofv3 runs of tools/collect_profiles.sh over `python bench.py` itself into the files committed under profiles/:
  r06_bench_kernel_stats.csv   per (kernel, grid size): launches, average / min / max duration -- from the kernel trace of ONE bench.py
                               process, whose own JSON line (HIP-event kernel_ms measured inside that process) is r06_bench_traced.json
  r06_bench_pmc.json           per kernel (the launch shape of the headline step = the shape with most launches): FETCH_SIZE / WRITE_SIZE per
                               launch and hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE doubled: gfx950 tallies the 128-byte
                               requests of wide coalesced reads at 64 bytes, MI355X_MICROARCH.md "HBM"; WRITE_SIZE as reported); "step" = the
                               kernels of one TableScan + JoinHash step added up, "hy_join_hash" = the join's.  bench.py reads this file for
                               roofline.traffic.
usage: summarize_bench_profile.py <dir with trace/ fetch/ write/> <commit>"""
import collections
import csv
import glob
import json
import os
import sys

STEP = ("scan_slices", "rank_table_fill_waves", "pk_count", "pk_scan", "pk_emit")   # (prepare_jobs and zero_vectors left the steady state in round 5)
JOIN = STEP[1:]


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("hy::", "").strip()


def base(name):
    return short(name).split("<")[0]


def timeline(root):
    """r06_bench_step_timeline.txt: the kernels of three consecutive headline steps in the middle of the timed region, with the idle time
    before each (launch gaps) -- what separates the sum of the kernels from the step's wall time."""
    rows = []
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(fh)]
    rows.sort()
    emits = [i for i, r in enumerate(rows) if base(r[2]) == "pk_emit"]
    scans = [i for i, r in enumerate(rows) if base(r[2]) == "scan_slices"]
    if len(emits) < 12 or len(scans) < 8:
        return
    # the timed region is where scan and join launches alternate (the joins in front of it -- placement calibration, setup -- come without
    # scans): three steps around the middle SCAN launch
    mid = len(scans) // 2
    steps_begin = scans[mid - 1]
    later = [i for i in emits if i > scans[min(len(scans) - 1, mid + 1)]]
    if not later:
        return
    end = later[0]
    with open(os.path.join(root, "r06_bench_step_timeline.txt"), "w") as fh:
        fh.write("offset_us  idle_before_us  duration_us  kernel\n")
        origin, previous_end = rows[steps_begin][0], None
        for start, stop, name in rows[steps_begin:end + 1]:
            idle = (start - previous_end) / 1e3 if previous_end is not None else 0.0
            fh.write(f"{(start - origin) / 1e3:9.1f}  {idle:14.1f}  {(stop - start) / 1e3:11.1f}  {name[:70]}\n")
            previous_end = max(stop, previous_end or stop)


def main():
    root, commit = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    groups = collections.defaultdict(list)
    launches = []
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "hy::" in row["Kernel_Name"]:
                    key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]))
                    groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                    launches.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), key))
    timeline(root)
    # The step's kernels INSIDE the stretch where scans and joins alternate (warm-up, the timed region and the scan-alone leg behind it): the
    # joins in front of it run into the calibration's candidate placements (bench.py --placements) and the join legs behind it into buffers
    # of their own -- the same kernels, slower or faster with where their output lies -- and would blur the step's averages.
    scans = sorted(start for start, _, key in launches if base(key[0]) == "scan_slices")
    in_step = collections.defaultdict(list)
    if scans:
        for start, stop, key in launches:
            if scans[0] <= start <= scans[-1] and base(key[0]) in STEP:
                in_step[key].append(stop - start)
    with open(os.path.join(root, "r06_bench_kernel_stats.csv"), "w", newline="") as fh:
        writer = csv.writer(fh)
        writer.writerow(["kernel", "grid_size_x", "launches", "average_us", "min_us", "max_us", "total_ms", "launches_in_step_region", "average_us_in_step_region"])
        for (name, grid), durations in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            inside = in_step.get((name, grid), [])
            writer.writerow([name, grid, len(durations), f"{sum(durations) / len(durations) / 1e3:.2f}", f"{min(durations) / 1e3:.2f}", f"{max(durations) / 1e3:.2f}", f"{sum(durations) / 1e6:.3f}",
                             len(inside), f"{sum(inside) / len(inside) / 1e3:.2f}" if inside else ""])
    # the headline shape of every kernel: the (name, grid) with most launches
    headline = {}
    for (name, grid), durations in groups.items():
        if base(name) not in headline or len(durations) > headline[base(name)][2]:
            headline[base(name)] = (name, grid, len(durations), sum(durations) / len(durations) / 1e3)
    for name in STEP:
        if name in headline:
            print(f"{headline[name][0]:40s} grid {headline[name][1]:9d} launches {headline[name][2]:5d} avg {headline[name][3]:8.1f} us")

    def counters(leg, counter):
        per = collections.defaultdict(lambda: [0.0, 0])
        for path in glob.glob(os.path.join(root, leg, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") == counter and "hy::" in row.get("Kernel_Name", ""):
                        cell = per[(short(row["Kernel_Name"]), int(row["Grid_Size"]) if "Grid_Size" in row else int(row.get("Grid_Size_X", 0)))]
                        cell[0] += float(row["Counter_Value"])
                        cell[1] += 1
        return per

    fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
    kernels = {}
    for name, (full, grid, launches, average) in headline.items():
        f, w = fetch.get((full, grid), [0.0, 0]), write.get((full, grid), [0.0, 0])
        if not f[1] and not w[1]:
            continue
        kernels[name] = {"kernel": full, "grid_size_x": grid, "launches_counted": max(f[1], w[1]), "FETCH_SIZE_KB_per_launch": f[0] / f[1] if f[1] else None,
                         "WRITE_SIZE_KB_per_launch": w[0] / w[1] if w[1] else None,
                         "hbm_bytes_per_launch": (f[0] / f[1] * 2048 if f[1] else 0) + (w[0] / w[1] * 1024 if w[1] else 0), "average_us_traced": average}
    if "sd_groups" in kernels:   # config 4: the two launches of the Q1-shaped AggregateHash (bench.py names them together)
        members = [m for m in ("sd_groups", "sd_wide") if m in kernels]
        kernels["sd_groups + sd_wide"] = {"kernels": members, "hbm_bytes_per_launch": sum(kernels[m]["hbm_bytes_per_launch"] for m in members),
                                          "average_us_traced": sum(kernels[m]["average_us_traced"] for m in members)}
    for total, members in (("step", STEP), ("hy_join_hash", JOIN)):
        if all(m in kernels for m in members if m not in ("prepare_jobs", "zero_vectors", "pk_plan")):
            kernels[total] = {"kernels": [m for m in members if m in kernels], "hbm_bytes_per_launch": sum(kernels[m]["hbm_bytes_per_launch"] for m in members if m in kernels)}
    summary = {"collected": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) over `python bench.py`, commit {commit}", "kernels": kernels,
               "note": "bytes = 2 x FETCH_SIZE KB x 1024 + WRITE_SIZE KB x 1024 per launch of the kernel's headline shape (the grid size with most launches); memory-side cache hits included"}
    with open(os.path.join(root, "r06_bench_pmc.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    for name in ("scan_slices", "pk_emit", "pk_count", "rank_table_fill_waves", "step", "hy_join_hash", "sd_groups", "sd_wide", "sd_groups + sd_wide"):
        if name in kernels:
            print(f"{name:28s} hbm bytes per launch {kernels[name]['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
