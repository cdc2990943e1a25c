# usage: bash tools/ab/table_ab.sh <outdir> lib...  : configuration table per library (default settings)
set -u
out=$1; shift
mkdir -p $out
for lib in "$@"; do
  tag=$(basename $lib .so)
  PT_LIBPTCORE=$PWD/$lib timeout -k 10 300 python3 tools/config_table.py 2>&1 | grep -v amdgpu.ids > $out/table_$tag.txt || { echo "failed: $tag"; tail -3 $out/table_$tag.txt; continue; }
  echo "== $tag"; cut -c1-38,84-120,170-212 $out/table_$tag.txt
done
