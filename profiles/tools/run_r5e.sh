#!/bin/bash
OUT=gpurun_out/r5e; mkdir -p $OUT
for R in 1 2; do for V in startup drv_lds tp_flat; do SOLVE_FLAGS=64 python profiles/tools/ab_timing.py profiles/tools/_ab/$V.so profiles/tools/_ab/$V.so S64_K30 2>&1 | grep solve_kernel | head -1; done; done > $OUT/ab11.txt
cat $OUT/ab11.txt
