#!/bin/bash
OUT=gpurun_out/r5b; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_time_parallel_oracle_gpu.py tests/test_time_parallel_gpu.py "tests/test_full_size_gpu.py::test_host_pointer_calls_have_no_stragglers" "tests/test_discretize_gpu.py::test_scipy_zoh_mode_vs_reference" "tests/test_mpc_loop_gpu.py::test_several_devices_from_the_api" -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -15 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 600 python profiles/tools/host_trace_stats.py 2000 > $OUT/host_trace_stats.txt 2>&1 || { tail -5 $OUT/host_trace_stats.txt; exit 1; }
cat $OUT/host_trace_stats.txt
