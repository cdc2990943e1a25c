#!/usr/bin/env python3
"""Copies what tools/measure_round.sh left under gpurun_out/<tag>/ into profiles/ under round-named
files: the benchmark lines, the configuration tables, the occupancy profile, the rocprofv3 kernel stats and
PMC summaries of the headline and the 20 M-triangle launch, and profiles/pmc_summary.json (read by
bench.py for its roofline block).

    python tools/collect_round.py <tag> <prefix>        e.g.  r02b r02
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
names = {"c2": "c2_dragonglass_1080p", "c1": "c1_bunnyclearcoat_512", "c3": "c3_interior_1080p", "c5": "c5_spheres_1080p",
         "c5_wavefront": "c5_spheres_16spp_wavefront", "c5_megakernel16": "c5_spheres_16spp_megakernel", "c4": "c4_20Mtri_4k_device_bvh", "c2_closeup": "c2_closeup_1080p"}
for k, n in names.items():
    line = open(os.path.join(src, f"bench_{k}.out")).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{prefix}_bench_{n}.json"), "w"), indent=1)
for k in ("config_table_default", "config_table_host", "config_table_device", "config_table_full-sweep", "occupancy"):
    text = "\n".join(l for l in open(os.path.join(src, k + ".out")).read().splitlines() if "amdgpu.ids" not in l)
    open(os.path.join(dst, f"{prefix}_{k}.txt"), "w").write(text + "\n")
py = sys.executable
for cfg, shape in (("c2", "scene=dragonglass width=1920 height=1080 spp_per_step=64 max_bounces=8 subdivide=1 bvh=host copies=1 rebuild=1"),
                   ("c2_closeup", "scene=closeup width=1920 height=1080 spp_per_step=64 max_bounces=8 subdivide=1 bvh=host copies=1 rebuild=1"),
                   ("c4", "scene=dragonglass width=3840 height=2160 spp_per_step=16 max_bounces=8 subdivide=0 bvh=device copies=139 rebuild=0")):
    prof = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{cfg}")
    pmc = os.path.join(dst, f"{prefix}_{cfg}_pmc.json")
    subprocess.run([py, os.path.join(ROOT, "tools", "summarize_pmc.py"), prof, pmc], check=True, stdout=subprocess.DEVNULL)
    t = json.load(open(pmc))
    t["source"] = f"gpurun_out/prof_{tag}_{cfg} (tools/profile_gpu.sh: rocprofv3 --kernel-trace --stats, then one --pmc run per counter group)"
    json.dump(t, open(pmc, "w"), indent=1, sort_keys=True)
    stats = glob.glob(os.path.join(prof, "kt", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f"{prefix}_{cfg}_kernel_stats.csv"))  # (the newest run's)
    out = os.path.join(dst, "pmc_summary.json" if cfg == "c2" else f"{prefix}_{cfg}_pmc_summary.json")
    subprocess.run([py, os.path.join(ROOT, "tools", "make_pmc_summary.py"), os.path.relpath(pmc, ROOT), out] + shape.split(),
                   check=True, stdout=subprocess.DEVNULL, cwd=ROOT)
print("collected", tag, "->", dst)
