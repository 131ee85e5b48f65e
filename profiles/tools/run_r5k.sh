#!/bin/bash
# per-phase cycles of solve_kernel under load (2048 satellites = two waves per SIMD everywhere), round-4 sources against HEAD's
OUT=gpurun_out/r5k; mkdir -p $OUT
SRC=$PWD/profiles/tools/_ab/tree_r4/mpconstellation_amd/csrc timeout -k 10 400 python profiles/tools/phase_timing.py 2048 > $OUT/phase_timing_load_r4src.txt 2>&1 || { tail -5 $OUT/phase_timing_load_r4src.txt; exit 1; }
timeout -k 10 400 python profiles/tools/phase_timing.py 2048 > $OUT/phase_timing_load_head.txt 2>&1 || { tail -5 $OUT/phase_timing_load_head.txt; exit 1; }
paste <(grep "per call" $OUT/phase_timing_load_r4src.txt | cut -c1-75) <(grep "per call" $OUT/phase_timing_load_head.txt | awk '{print $NF}')
grep "lived" $OUT/phase_timing_load_r4src.txt $OUT/phase_timing_load_head.txt | cut -c1-200
