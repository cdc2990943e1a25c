#!/bin/bash
# Runs on the GPU box (from the repo root): the round's benchmark lines of the five configurations,
# the scheduler A/B on the sphere field, the configuration table and the rocprofv3 passes of the
# headline launch and of the 20 M-triangle launch.  Everything lands under gpurun_out/<tag>/.
#   tools/measure_round.sh <tag>
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
step() {  # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"
  timeout -k 10 "$secs" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" || { echo "$name failed (rc $?)"; tail -5 "$OUT/$name.err"; exit 1; }
}
C4="--copies 139 --width 3840 --height 2160 --bvh device --steps 2 --warmup 1 --spp-per-step 16"
step bench_c2 400 python3 bench.py
step bench_c1 300 python3 bench.py --scene c1
step bench_c3 400 python3 bench.py --scene interior
step bench_c5 400 python3 bench.py --scene spheres
step bench_c2_closeup 400 python3 bench.py --scene closeup --no-cpu-baseline
step bench_c5_wavefront 400 python3 bench.py --scene spheres --tuning pipeline=1 --no-cpu-baseline --no-device-tree --no-frame-by-frame --steps 2 --spp-per-step 16
step bench_c5_megakernel16 400 python3 bench.py --scene spheres --no-cpu-baseline --no-device-tree --steps 2 --spp-per-step 16
step bench_c4 600 python3 bench.py $C4 --no-cpu-baseline
step config_table_default 400 python3 tools/config_table.py --depths
step config_table_host 400 python3 tools/config_table.py --no-rebuild
step config_table_device 400 python3 tools/config_table.py --bvh device
step config_table_full-sweep 500 python3 tools/config_table.py --bvh full-sweep --no-rebuild
step occupancy 200 python3 tools/occupancy_profile.py 64
echo "== profiles"
bash tools/profile_gpu.sh ${TAG}_c2 > "$OUT/profile_c2.log" 2>&1 || { echo "profile c2 failed"; tail -5 "$OUT/profile_c2.log"; exit 1; }
bash tools/profile_gpu.sh ${TAG}_c2_closeup --scene closeup --steps 2 --warmup 1 --no-cpu-baseline --no-device-tree --no-frame-by-frame > "$OUT/profile_c2_closeup.log" 2>&1 || { echo "profile close-up failed"; tail -5 "$OUT/profile_c2_closeup.log"; exit 1; }
bash tools/profile_gpu.sh ${TAG}_c4 $C4 --no-cpu-baseline --no-frame-by-frame > "$OUT/profile_c4.log" 2>&1 || { echo "profile c4 failed"; tail -5 "$OUT/profile_c4.log"; exit 1; }
echo "measure $TAG done"
