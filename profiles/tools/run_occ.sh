#!/bin/bash
# occupancy sweep of solve_kernel: the same kernel with unused LDS added so that 7, 6, 5, 4, 3 workgroups fit a compute unit instead of 8
A=profiles/tools/_ab
for V in head_r5 occ7 occ6 occ5 occ4 occ3; do
  timeout -k 10 200 python profiles/tools/ab_timing.py $A/$V.so $A/$V.so S4096_K30 S8192_K30 2>&1 | grep solve_kernel | head -2
done
