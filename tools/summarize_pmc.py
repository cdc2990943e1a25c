#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs that tools/profile_gpu.sh wrote into one JSON:
per-kernel average duration (kernel-trace) and per-launch counter means (PMC passes).

    python tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<name>.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def newest_per_pass(paths):
    """One file per pass directory, the newest: gpurun merges a run's output into what earlier runs with the
    same tag left behind, under other process ids."""
    best = {}
    for path in paths:
        d = os.path.dirname(path)
        if d not in best or os.path.getmtime(path) > os.path.getmtime(best[d]):
            best[d] = path
    return sorted(best.values())


def main(src, dst):
    out = {"source": src, "kernels": {}}
    for path in newest_per_pass(glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            name = row["Name"].split("(")[0]
            out["kernels"][name] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                    "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                                    "percent": float(row["Percentage"])}
    counters = defaultdict(lambda: defaultdict(list))
    meta = {}
    for path in newest_per_pass(glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            counters[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                   "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for name, cs in counters.items():
        k = out["kernels"].setdefault(name, {})
        k["launch_info"] = meta.get(name)
        k["counters_mean_per_launch"] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        k["counter_samples"] = {c: len(v) for c, v in sorted(cs.items())}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
