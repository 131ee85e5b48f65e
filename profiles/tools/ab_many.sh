#!/bin/bash
# A/B of several builds of libmpcx.so against a base on one box: bit-identity of the results, then alternating timings
# usage: profiles/tools/ab_many.sh base.so v1.so [v2.so ...]   (run from the repo root on the GPU box)
BASE=$1; shift
for V in "$@"; do
  echo "== compare $(basename $BASE) $(basename $V)"
  python profiles/tools/ab_compare.py "$BASE" "$V" 2>&1 | grep -v amdgpu.ids | tail -12
done
for R in 1 2; do
  for V in "$BASE" "$@"; do
    python profiles/tools/ab_timing.py "$V" "$V" ${WL:-S64_K30 S4096_K30} 2>&1 | grep solve_kernel | head -${NL:-2}
  done
done
