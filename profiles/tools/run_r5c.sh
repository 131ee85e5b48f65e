#!/bin/bash
OUT=gpurun_out/r5c; mkdir -p $OUT
MPCX_HOST_TRACE=10 timeout -k 10 600 python profiles/tools/host_trace_stats.py 8000 > $OUT/host_trace_stats.txt 2> $OUT/host_trace_slow.txt || { tail -5 $OUT/host_trace_stats.txt; exit 1; }
cat $OUT/host_trace_stats.txt; grep -c "host trace" $OUT/host_trace_slow.txt; head -c 6000 $OUT/host_trace_slow.txt
nproc; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null; uptime
