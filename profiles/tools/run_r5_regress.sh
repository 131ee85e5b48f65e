#!/bin/bash
# end-of-round regression of the default kernels on the sweeps of rounds 3/4 (convergence counts and iteration statistics): into gpurun_out/r5reg/
O=gpurun_out/r5reg; mkdir -p $O
timeout -k 10 400 python profiles/tools/scenario_sweep.py > $O/scenario_sweep.txt 2>&1 && echo scenario done && \
timeout -k 10 400 python profiles/tools/mpc_option_sweep.py > $O/mpc_option_sweep.txt 2>&1 && echo option done && \
timeout -k 10 300 python profiles/tools/edge_cases.py > $O/edge_cases.txt 2>&1 && echo edge done
for f in $O/*.txt; do tail -n 3 $f; done
