#!/usr/bin/env python3
"""Register, spill, scratch and LDS figures of every kernel of libptcore.so, one line each, from the
compiler's -Rpass-analysis=kernel-resource-usage remarks (`make -C pathtracer_amd/csrc resource-usage`).

    python tools/resource_usage.py [substring of the kernel name ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "pathtracer_amd", "csrc"), "resource-usage"],
                     capture_output=True, text=True).stderr
demangle = {}
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    key, val = m.group(1), m.group(2)
    if key == "Function Name":
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    short = re.sub(r"\(.*", "", n).replace("void ptcore::", "").replace("ptcore::", "")
    if sys.argv[1:] and not any(a in short for a in sys.argv[1:]):
        continue
    print(f"{short:46s} VGPRs {r.get('VGPRs', '?'):>4s}  spilled {r.get('VGPRs Spill', '?'):>4s}  scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} B/lane  "
          f"SGPRs {r.get('TotalSGPRs', '?'):>4s} (spilled {r.get('SGPRs Spill', '?'):>4s})  LDS {r.get('LDS Size [bytes/block]', '?'):>6s} B  "
          f"waves/SIMD {r.get('Occupancy [waves/SIMD]', '?')}")
