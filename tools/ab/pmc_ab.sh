# usage: [PT_AB_ARGS="--scene interior"] bash tools/ab/pmc_ab.sh <outdir> lib...   : SQ instruction counters of the benchmark launch per library
set -u
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
ARGS="${PT_AB_ARGS:-} --steps 2 --warmup 1 --no-cpu-baseline --no-device-tree --no-frame-by-frame"
for lib in "$@"; do
  tag=$(basename $lib .so)
  PT_LIBPTCORE=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/$tag -- python3 bench.py $ARGS > $out/$tag.log 2>&1 || { echo "pass failed for $tag"; tail -3 $out/$tag.log; exit 1; }
  PT_LIBPTCORE=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/${tag}_b -- python3 bench.py $ARGS > $out/${tag}_b.log 2>&1 || { echo "pass b failed for $tag"; exit 1; }
done
python3 - "$out" "$@" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
for lib in sys.argv[2:]:
    tag=os.path.basename(lib)[:-3]
    c=defaultdict(list)
    for d in (tag,tag+'_b'):
        for path in glob.glob(os.path.join(out,d,'**','*counter_collection.csv'),recursive=True):
            for row in csv.DictReader(open(path)):
                if 'renderKernel' in row['Kernel_Name'] and 'false, 3, true' in row['Kernel_Name']:
                    c[row['Counter_Name']].append(float(row['Counter_Value']))
    # the last launches are the 64-spp steps; take the largest values' mean (probe launches are small)
    print(tag, {k:'%.4g'%(sum(sorted(v)[-2:])/2) for k,v in sorted(c.items())})
PY
