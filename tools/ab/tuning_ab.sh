# usage: bash tools/ab/tuning_ab.sh <outfile> "<bench args>" tuning1 tuning2 ...   (one bench.py line per --tuning value, "-" = none)
out=$1; args=$2; shift 2
mkdir -p $(dirname $out)
: > $out
for tn in "$@"; do
  t=""; [ "$tn" != "-" ] && t="--tuning $tn"
  timeout -k 10 400 python3 bench.py $args $t --no-cpu-baseline --no-device-tree --no-frame-by-frame 2>/dev/null | tail -1 | python3 -c "
import json,sys
t=json.loads(sys.stdin.read()); r=t['roofline']
print('%-28s %-22s %9.1f Mrays/s  %8.3f ms/step (median %8.3f)  kernel %8.3f ms' % ('$tn', '$args', t['value'], t['ms_per_step'], t['ms_per_step_median'], r['kernel_ms_avg']))" >> $out || exit 1
done
cat $out
