#!/bin/bash
OUT=gpurun_out/r5d; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_time_parallel_oracle_gpu.py "tests/test_full_size_gpu.py::test_host_pointer_calls_have_no_stragglers" "tests/test_mpc_loop_gpu.py" -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -8 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 600 python profiles/tools/update_split.py 64 2048 4096 8192 > $OUT/update_split.txt 2>&1 || { tail -5 $OUT/update_split.txt; exit 1; }
cat $OUT/update_split.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/cl -- python3 $GRAFT_REPO_ROOT/profiles/tools/closed_loop_trace.py 4096 > $GRAFT_REPO_ROOT/$OUT/cl.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$OUT/cl.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 profiles/tools/closed_loop_trace.py --summarize $OUT/cl > $OUT/closed_loop_trace.txt 2>&1; cat $OUT/closed_loop_trace.txt | head -60
