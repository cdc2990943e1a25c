#!/bin/bash
# Runs on the GPU box (from the repo root): kernel-trace stats + PMC passes of the benchmark
# command, one rocprofv3 run per pass.  Results land under gpurun_out/prof_<tag>/.
#   tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ARGS=${*:-"--steps 2 --warmup 1 --no-cpu-baseline --no-device-tree --no-frame-by-frame"}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass tcc1 FETCH_SIZE TCC_HIT_sum
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
echo "profile $TAG done"
