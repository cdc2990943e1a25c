# like bench_ab.sh, and prints the resolve kernel's time as well
out=$1; args=$2; shift 2
mkdir -p $(dirname $out)
: > $out
for lib in "$@"; do
  PT_LIBPTCORE=$PWD/$lib timeout -k 10 400 python3 bench.py $args --no-cpu-baseline --no-device-tree --no-frame-by-frame 2>/dev/null | tail -1 | python3 -c "
import json,sys
t=json.loads(sys.stdin.read()); r=t['roofline']
print('%-24s %-20s %9.1f Mrays/s  %8.3f ms/step  kernel %8.3f ms  resolve %6.3f ms' % ('$lib', '$args', t['value'], t['ms_per_step'], r['kernel_ms_avg'], r['resolve_kernel']['kernel_ms_avg']))" >> $out || exit 1
done
cat $out
