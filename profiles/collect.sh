#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/<round>/ (run on the GPU box from the repo root):
#   1. kernel trace + stats per workload (kernel durations the bench line's roofline.achieved must agree with)
#   2. FETCH_SIZE and WRITE_SIZE in separate counter passes (TCC slots do not hold both), kernel trace only
#   3. two passes of 8 SQ counters each: instruction mix / fp64 operation counts, and wait / active cycles
# Raw output goes to gpurun_out/prof/<round>/ ; profiles/summarize.py turns it into the committed summaries.
set -e
ROUND=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in S4096_K30 S64_K30 S4096_K100_scp2 S8192_K30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/stats" -o stats -- \
    python3 "$REPO/bench.py" --workload $W --steps 8 --warmup 2 --no-also --no-cpu-baseline > "$OUT/$W.bench.log" 2>&1
  echo "stats $W done"
done
# the time-parallel kernel (MPCX_SOLVE_TIME_PARALLEL) on configs[1]: kernel durations only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/S64_K30_tp/stats" -o stats -- \
  python3 "$REPO/bench.py" --workload S64_K30 --solve-flags 64 --steps 8 --warmup 2 --no-also --no-cpu-baseline > "$OUT/S64_K30_tp.bench.log" 2>&1 \
  || echo "(the profiled process of the cooperative launch ends with a fault in its exit handlers, after the bench line and the profiler's files are written: seen under rocprofv3 only)"
echo "stats S64_K30_tp done"
SQ_A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"
SQ_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F64"
# (K = 100: the configuration BASELINE.json calls the HBM-bound regime)
for W in S4096_K30 S64_K30 S4096_K100_scp2; do
  for C in FETCH_SIZE WRITE_SIZE SQ_A SQ_B; do
    case $C in SQ_A) LIST=$SQ_A;; SQ_B) LIST=$SQ_B;; *) LIST=$C;; esac
    rocprofv3 --kernel-trace --pmc $LIST --output-format csv -d "$OUT/$W/$C" -o pmc -- \
      python3 "$REPO/bench.py" --workload $W --steps 1 --warmup 1 --no-also --no-cpu-baseline > "$OUT/$W.$C.log" 2>&1
    echo "pmc $C $W done"
  done
done
# the time-parallel kernel's counters (round-4 verdict: its bench line had no traffic figure).  Last, each pass under its own
# time limit: the profiled process of a cooperative launch has been seen to fault in its exit handlers under rocprofv3.
for C in FETCH_SIZE WRITE_SIZE SQ_A SQ_B; do
  case $C in SQ_A) LIST=$SQ_A;; SQ_B) LIST=$SQ_B;; *) LIST=$C;; esac
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $LIST --output-format csv -d "$OUT/S64_K30_tp/$C" -o pmc -- \
    python3 "$REPO/bench.py" --workload S64_K30 --solve-flags 64 --steps 1 --warmup 1 --no-also --no-cpu-baseline > "$OUT/S64_K30_tp.$C.log" 2>&1 \
    || echo "(pmc $C S64_K30_tp: profiled process ended with an error after the run -- see the log)"
  echo "pmc $C S64_K30_tp done"
done
