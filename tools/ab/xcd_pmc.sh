# L2 hit rate and kernel time of the headline launch with one work queue and with eight (one per group of waves under one L2)
set -u
out=$1
mkdir -p $out
export TMPDIR=/tmp
for t in xcd_queues=0 xcd_queues=1; do
  ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-device-tree --no-frame-by-frame --tuning $t"
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_REQ_sum --output-format csv -d $out/$t -- python3 bench.py $ARGS > $out/$t.json 2> $out/$t.err || { echo fail; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/${t}_b -- python3 bench.py $ARGS > /dev/null 2>&1 || { echo fail b; exit 1; }
done
python3 - $out <<'PY'
import csv,glob,sys,os,json
from collections import defaultdict
out=sys.argv[1]
for t in ("xcd_queues=0","xcd_queues=1"):
    c=defaultdict(list)
    for d in (t,t+"_b"):
        for path in glob.glob(os.path.join(out,d,'**','*counter_collection.csv'),recursive=True):
            for row in csv.DictReader(open(path)):
                if 'renderKernel' in row['Kernel_Name']:
                    c[row['Counter_Name']].append(float(row['Counter_Value']))
    m={k:sum(sorted(v)[-3:])/3 for k,v in c.items()}
    line=[l for l in open(os.path.join(out,t+'.json')) if l.startswith('{')][-1]
    d=json.loads(line)
    print(t, 'kernel_ms %.2f  Mrays/s %.0f  L2 hit %.3f  L2 requests %.3g  fetch GB %.1f  L1->L2 reads %.3g  mean L1->L2 read latency %.0f cycles' % (d['roofline']['kernel_ms_avg'], d['value'], m['TCC_HIT_sum']/m['TCC_REQ_sum'], m['TCC_REQ_sum'], 2*m.get('FETCH_SIZE',0)*1024/1e9, m.get('TCP_TCC_READ_REQ_sum',0), m.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(m.get('TCP_TCC_READ_REQ_sum',1),1)))
PY
