#!/bin/bash
# the round-end check in one gpurun call: the whole GPU suite, then the default bench line (RUN=<name>: output under gpurun_out/<name>/)
OUT=gpurun_out/${RUN:-r5n}; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest.log 2>&1; rc=$?
tail -16 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 900 python bench.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
head -c 1800 $OUT/bench.log; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/self/cgroup | head -3
