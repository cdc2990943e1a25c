#!/usr/bin/env python3
"""Counter-derived figures of the trace kernel, for bench.py's roofline block, from the JSON that
tools/summarize_pmc.py wrote of a tools/profile_gpu.sh run of the benchmark command.

    python tools/make_pmc_summary.py profiles/<run>_pmc.json profiles/pmc_summary.json [key=value launch shape ...]

The launch shape (scene, width, height, spp_per_step, max_bounces, subdivide, bvh, copies) is stored with
the figures: bench.py uses them only for a launch of the same shape.
"""
import json
import sys

SIMDS = 256 * 4          # SIMDs of the chip
WAVE = 64


def main(src, dst, shape):
    t = json.load(open(src))
    name = max((k for k in t["kernels"] if "renderKernel" in k and "counters_mean_per_launch" in t["kernels"][k]
                and "true" not in k.split("<")[1].split(",")[2]),
               key=lambda k: t["kernels"][k].get("avg_ns", 0))
    k = t["kernels"][name]
    c = k["counters_mean_per_launch"]
    fetch, write = c["FETCH_SIZE"], c["WRITE_SIZE"]  # KB
    traffic = (2.0 * fetch + write) * 1024.0
    busy = c["GRBM_GUI_ACTIVE"] / 8.0  # cycles the chip was busy during the launch (the counter sums the eight XCDs)
    out = dict(shape)
    out.update({
        "source": src,
        "kernel": name.replace("void ", ""),
        "kernel_ms_avg": k["avg_ns"] / 1e6,
        "bytes_per_launch": traffic,
        "fetch_size_kb": fetch,
        "write_size_kb": write,
        "correction": ("gfx950: FETCH_SIZE tallies 128-B fabric requests as 64 B, so it is doubled; WRITE_SIZE is taken as reported; "
                       "both are in KB.  The counters sit on the L2's fabric side: Infinity-Cache hits are included."),
        # share of the launch's cycles in which a SIMD's VALU was executing (4 cycles per wave instruction)
        # (at most 1: a wave instruction whose active lanes all sit in one half of the wave issues in two cycles, so a
        # divergent kernel can retire more than one instruction per four cycles -- valu_cycles_per_inst says how many)
        "valu_busy": min(1.0, 4.0 * c["SQ_ACTIVE_INST_VALU"] / (busy * SIMDS)) if "SQ_ACTIVE_INST_VALU" in c else None,
        "valu_cycles_per_inst": busy * SIMDS / c["SQ_INSTS_VALU"],
        "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (WAVE * c["SQ_INSTS_VALU"]),
        # SQ_WAVE_CYCLES counts resident waves in units of four cycles
        "waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / (busy * SIMDS),
        "wave_wait_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "l2_hit": c["TCC_HIT_sum"] / max(c["TCC_REQ_sum"], 1.0),
        "valu_insts": c["SQ_INSTS_VALU"], "salu_insts": c["SQ_INSTS_SALU"], "vmem_rd_insts": c["SQ_INSTS_VMEM_RD"],
        "launch_info": k.get("launch_info"),
    })
    vb = out["valu_busy"] or 0.0
    out["bound"] = "valu-issue" if vb >= 0.6 else ("hbm" if traffic / (k["avg_ns"] / 1e9) / 1e9 > 4000.0 else "latency")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    shape = {}
    for item in sys.argv[3:]:
        key, val = item.split("=")
        shape[key] = val if key in ("scene", "bvh") else int(val)
    main(sys.argv[1], sys.argv[2], shape)
