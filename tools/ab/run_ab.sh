# usage: bash tools/ab/run_ab.sh <outfile> lib1 lib2 ...   (headline workload on the reference tree and on the device tree)
out=$1; shift
mkdir -p $(dirname $out)
timeout -k 10 400 python3 tools/ab_libs.py "$@" > $out 2>&1 && timeout -k 10 400 python3 tools/ab_libs.py --bvh device "$@" >> $out 2>&1
grep -v amdgpu.ids $out
