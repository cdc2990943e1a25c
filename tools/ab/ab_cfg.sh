# usage: bash tools/ab/ab_cfg.sh <outdir> lib...   (configuration table + 20 M-triangle line per library)
set -u
out=$1; shift
mkdir -p $out
for lib in "$@"; do
  tag=$(basename $lib .so)
  PT_LIBPTCORE=$PWD/$lib timeout -k 10 300 python3 tools/config_table.py 2>&1 | grep -v amdgpu.ids > $out/table_$tag.txt || exit 1
  PT_LIBPTCORE=$PWD/$lib timeout -k 10 300 python3 bench.py --copies 139 --width 3840 --height 2160 --bvh device --steps 2 --warmup 1 --spp-per-step 16 --no-cpu-baseline --no-frame-by-frame > $out/c4_$tag.json 2> $out/c4_$tag.err || exit 1
  echo "== $tag"; cut -c1-38,84-140,170-215 $out/table_$tag.txt
  python3 -c "
import json;d=json.load(open('$out/c4_$tag.json'));print('C4 %.0f Mrays/s %.1f ms  %.2f visits %.2f tests' % (d['value'],d['ms_per_step'],d['roofline']['node_visits_per_ray'],d['roofline']['triangle_tests_per_ray']))"
done
