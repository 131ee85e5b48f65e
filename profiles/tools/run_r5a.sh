#!/bin/bash
# round 5, first GPU call: the GPU suite, the host overhead of a multi-device call before / after the in-place result set, the bench
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 300 python profiles/tools/devices8_overhead.py profiles/tools/_ab/pkg_r04 > $OUT/dev8_before.txt 2>&1 || { tail -5 $OUT/dev8_before.txt; exit 1; }
tail -3 $OUT/dev8_before.txt
timeout -k 10 300 python profiles/tools/devices8_overhead.py > $OUT/dev8_after.txt 2>&1 || { tail -5 $OUT/dev8_after.txt; exit 1; }
tail -3 $OUT/dev8_after.txt
timeout -k 10 600 python bench.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
tail -c 1500 $OUT/bench.log
