#!/bin/bash
# rates of the encodings beside the headline, then the same run under rocprofv3 for the per-kernel table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02m
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 python3 -u $R/tools/misc_rates.py > $O/rates.log 2>&1 || { tail -20 $O/rates.log; exit 1; }
cat $O/rates.log
timeout -k 10 250 rocprofv3 --kernel-trace -d $O/prof -o v -- python3 $R/tools/misc_rates.py > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 $R/tools/kernel_times.py $O/prof/v_results.db > $O/kernels.txt 2>&1
cat $O/kernels.txt
