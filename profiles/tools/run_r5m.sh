#!/bin/bash
OUT=gpurun_out/r5m; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
exit $rc
