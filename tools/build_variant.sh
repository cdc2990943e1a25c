#!/bin/bash
# Builds a variant of libptcore.so for A/B measurements (tools/ab_libs.py, PT_LIBPTCORE):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DPT_QUANT_MARGIN=0]   ->  tools/ab/<name>.so
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pathtracer_amd/csrc
mkdir -p "$ROOT/tools/ab"
make -s -C "$SRC" host_bvh.o host_env.o host_camera.o host_pack.o host_gltf.o host_cache.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread --offload-arch=gfx950 \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -c "$SRC/ptcore.hip" -o "/tmp/ptcore_$NAME.o"
/opt/rocm/bin/hipcc -shared -fPIC -pthread -o "$ROOT/tools/ab/$NAME.so" "/tmp/ptcore_$NAME.o" "$SRC"/host_*.o
echo "built tools/ab/$NAME.so"
