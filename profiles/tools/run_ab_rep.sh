#!/bin/bash
# repeated alternating timings of builds (no compare): usage run_ab_rep.sh ROUNDS lib1.so lib2.so ...   (WL: workloads)
R=$1; shift
for i in $(seq 1 $R); do
  for V in "$@"; do
    python profiles/tools/ab_timing.py $V $V ${WL:-S4096_K30 S8192_K30} 2>&1 | grep solve_kernel | head -${NL:-2} | tr '\n' ' '; echo
  done
done
