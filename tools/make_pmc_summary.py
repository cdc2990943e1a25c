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
        # SQ_ACTIVE_INST_VALU counts, in units of four cycles, the time waves spend with a vector instruction in issue;
        # against the SIMDs' cycles it says how many waves per SIMD do so on average (capped at 1).  The pipe itself
        # retires a wave instruction in about 2.4 cycles once two or more waves feed it (tools/ubench/halfwave.hip), so
        # valu_pipe_share = 2.4 / valu_cycles_per_inst is the share of the pipe's throughput in use.
        "valu_busy": min(1.0, 4.0 * c["SQ_ACTIVE_INST_VALU"] / (busy * SIMDS)) if "SQ_ACTIVE_INST_VALU" in c else None,
        "valu_busy_raw": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (busy * SIMDS) if "SQ_ACTIVE_INST_VALU" in c else None,
        "valu_pipe_share": min(1.0, 2.4 * c["SQ_INSTS_VALU"] / (busy * SIMDS)),
        # The vector L1 (TCP) takes one access per cycle and CU (an access is 8 bytes of a lane's scattered read, 16 of a
        # whole line read by eight lanes, up to 64 of adjacent 4- and 8-byte reads: tools/ubench/l1rate.hip under the
        # same counter); l1_busy is the share of the CUs' cycles with an access, l1_stall the share it spent held up
        # by misses in flight.
        "l1_busy": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (busy * 256.0) if "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None,
        "l1_stall": c["TCP_PENDING_STALL_CYCLES_sum"] / (busy * 256.0) if "TCP_PENDING_STALL_CYCLES_sum" in c else None,
        "l1_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "valu_cycles_per_inst": busy * SIMDS / c["SQ_INSTS_VALU"],
        "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (WAVE * c["SQ_INSTS_VALU"]),
        # SQ_WAVE_CYCLES counts resident waves in units of four cycles
        "waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / (busy * SIMDS),
        "wave_wait_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "l2_hit": c["TCC_HIT_sum"] / max(c["TCC_REQ_sum"], 1.0),
        # what the L2s were asked for: requests x the 128-byte line (an upper estimate, a request moves at most one
        # line), and how long a vector L1 waited for a line it asked the L2 for, on average, in L1 clock cycles
        "l2_requests": c["TCC_REQ_sum"],
        "l2_request_gbps": c["TCC_REQ_sum"] * 128.0 / (k["avg_ns"] / 1e9) / 1e9,
        "l1_l2_read_latency_cycles": (c["TCP_TCC_READ_REQ_LATENCY_sum"] / max(c["TCP_TCC_READ_REQ_sum"], 1.0)
                                      if "TCP_TCC_READ_REQ_LATENCY_sum" in c else None),
        "valu_insts": c["SQ_INSTS_VALU"], "salu_insts": c["SQ_INSTS_SALU"], "vmem_rd_insts": c["SQ_INSTS_VMEM_RD"],
        "launch_info": k.get("launch_info"),
    })
    # What limits the kernel.  (l1_busy + l1_stall is 1 whatever the kernel does -- stall cycles fill the cycles without an
    # access -- so the pair says nothing about the bound; probes with extra instructions and extra reads per node visit
    # show both the vector pipe and the L1 with slack: profiles/r03_bound_experiment.txt.)  A launch that moves more than
    # 4.8 TB/s through the fabric -- 0.6 of the HBM peak, 85 % of the 5.5-5.8 TB/s at which random rows of a buffer far
    # larger than the Infinity Cache can be gathered -- is called memory bound (the 20 M-triangle launch: 5.2 TB/s);
    # otherwise (the cache-resident scenes: 1.9-4.0 TB/s out of an Infinity Cache that delivers 8.6) it is a wave's serial
    # chain of one memory round trip and a few hundred vector instructions per traversal step, with the two other waves
    # of its SIMD to cover it.
    out["bound"] = "hbm"
    out["limited_by"] = ("fabric / HBM bandwidth" if traffic / (k["avg_ns"] / 1e9) / 1e9 > 4800.0 else
                         "latency and issue at 3 waves per SIMD (registers and LDS cap the occupancy)")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    shape = {}
    for item in sys.argv[3:]:
        key, val = item.split("=")
        shape[key] = val if key in ("scene", "bvh") else int(val)
    main(sys.argv[1], sys.argv[2], shape)
