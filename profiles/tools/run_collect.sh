#!/bin/bash
# the round's rocprofv3 evidence in one gpurun call: profiles/collect.sh, then the summaries (profiles/summarize.py) into profiles/<round>/
R=${1:-r05}
bash profiles/collect.sh $R > gpurun_out/collect_$R.log 2>&1; rc=$?
tail -30 gpurun_out/collect_$R.log
python3 profiles/summarize.py $R > gpurun_out/summarize_$R.log 2>&1
tail -5 gpurun_out/summarize_$R.log
mkdir -p gpurun_out/profiles_$R && cp -r profiles/$R/* gpurun_out/profiles_$R/
ls gpurun_out/profiles_$R
exit $rc
