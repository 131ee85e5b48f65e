#!/bin/bash
# counter passes on the headline workload: vector-memory pipe and L2 (raw csv under gpurun_out/pmc_mem/, sums / means per kernel printed)
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_mem; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=${1:-S4096_K30}
declare -A SETS
SETS[DER1]="MemUnitBusy MemUnitStalled"
SETS[DER2]="VALUBusy SALUBusy L2CacheHit"
SETS[TA]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
SETS[TCP]="TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
SETS[TCC]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_WRREQ_STALL_sum TCC_BUSY_avr"
for C in DER1 DER2 TCC; do      # (TA and TCP: rocprofv3 aborted at the time limit with these sets on this pool -- left out)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${SETS[$C]} --output-format csv -d "$OUT/$C" -o pmc -- \
    python3 "$REPO/bench.py" --workload $W --steps 1 --warmup 1 --no-also --no-cpu-baseline > "$OUT/$C.log" 2>&1 || { echo "pass $C failed"; tail -3 "$OUT/$C.log"; }
  echo "pmc $C done"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_mem/*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    print(f)
    for k, v in acc.items():
        if 'solve' in k or 'discretize' in k:
            print('  ', k[:50], {c: (round(sum(x) / len(x), 3), len(x)) for c, x in v.items()})
PY
