#!/bin/bash
# extra counter passes on the headline workload (instruction cache, wait reasons): raw csv under gpurun_out/pmc_extra/, sums per kernel printed
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_extra; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=${1:-S4096_K30}
declare -A SETS
SETS[IC]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL"
SETS[WAITS]="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAVES"
SETS[MISC]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"
for C in IC WAITS MISC; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${SETS[$C]} --output-format csv -d "$OUT/$C" -o pmc -- \
    python3 "$REPO/bench.py" --workload $W --steps 1 --warmup 1 --no-also --no-cpu-baseline > "$OUT/$C.log" 2>&1 || { echo "pass $C failed"; tail -5 "$OUT/$C.log"; }
  echo "pmc $C done"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_extra/*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    print(f)
    for k, v in acc.items():
        if 'solve' in k or 'discretize' in k:
            print('  ', k[:60], {c: x for c, x in v.items()})
PY
