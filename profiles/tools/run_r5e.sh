#!/bin/bash
OUT=gpurun_out/r5e; mkdir -p $OUT
WL="S4096_K30 S4096_K100_scp2 S8192_K30" bash profiles/tools/ab_many.sh profiles/tools/_ab/batched_loads.so profiles/tools/_ab/nosync.so > $OUT/ab4.txt 2>&1
cat $OUT/ab4.txt
