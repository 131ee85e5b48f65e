#!/bin/bash
OUT=gpurun_out/r5e; mkdir -p $OUT
WL="S4096_K30 S4096_K100_scp2" bash profiles/tools/ab_many.sh profiles/tools/_ab/batched_loads.so profiles/tools/_ab/pads.so > $OUT/ab3.txt 2>&1
cat $OUT/ab3.txt
