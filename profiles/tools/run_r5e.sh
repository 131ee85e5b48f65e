#!/bin/bash
OUT=gpurun_out/r5e; mkdir -p $OUT
WL="S4096_K30 S64_K30" bash profiles/tools/ab_many.sh profiles/tools/_ab/refine_path.so profiles/tools/_ab/startup.so > $OUT/ab6.txt 2>&1
cat $OUT/ab6.txt
