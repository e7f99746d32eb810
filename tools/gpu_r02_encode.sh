#!/bin/bash
# encoder rates, then the kernels' own times under rocprofv3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02enc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 -u $R/tools/encode_rates.py > $O/rates.log 2>&1 || { tail -20 $O/rates.log; exit 1; }
grep -v amdgpu.ids $O/rates.log
rm -rf "$O/prof"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -o v -- python3 $R/tools/encode_rates.py > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 $R/tools/kernel_times.py $O/prof/v_results.db 2>&1 | grep -E "kernel  |encode|bounds" > $O/kernels.txt
cat $O/kernels.txt
