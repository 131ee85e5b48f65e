#!/bin/bash
OUT=gpurun_out/r5f; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 900 python bench.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
head -c 2500 $OUT/bench.log
