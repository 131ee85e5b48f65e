#!/bin/bash
# the time-parallel kernel's sweeps against the default kernels, re-run on the round-5 kernels (light release with the vmcnt wait)
OUT=gpurun_out/r5i; mkdir -p $OUT
timeout -k 10 400 python profiles/tools/tp_option_sweep.py > $OUT/tp_option_sweep.txt 2>&1 || { tail -3 $OUT/tp_option_sweep.txt; exit 1; }
tail -4 $OUT/tp_option_sweep.txt
timeout -k 10 500 python profiles/tools/tp_fuzz.py 0 60 > $OUT/tp_fuzz.txt 2>&1 || { tail -3 $OUT/tp_fuzz.txt; exit 1; }
tail -2 $OUT/tp_fuzz.txt
timeout -k 10 600 python profiles/tools/tp_scenario_sweep.py > $OUT/tp_scenario_sweep.txt 2>&1 || { tail -3 $OUT/tp_scenario_sweep.txt; exit 1; }
tail -3 $OUT/tp_scenario_sweep.txt
