#!/bin/bash
# the time-parallel kernel's tests against a build with the agent-scope fences forced (-DMPCX_TP_HEAVY): round-4 verdict item 1a
OUT=gpurun_out/r5h; mkdir -p $OUT
MPCX_LIB=$PWD/profiles/tools/_ab/tp_heavy.so timeout -k 10 600 python -m pytest tests/test_time_parallel_oracle_gpu.py tests/test_time_parallel_gpu.py -m gpu -q -s > $OUT/tp_heavy_pytest.log 2>&1; rc=$?
grep -v "amdgpu.ids" $OUT/tp_heavy_pytest.log | tail -40
exit $rc
