#!/bin/bash
# rates of the lines of tools/misc_rates.py matching $1, then their kernels under rocprofv3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02one
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1"
timeout -k 10 200 python3 -u $R/tools/misc_rates.py > $O/rates.log 2>&1 || { tail -20 $O/rates.log; exit 1; }
grep "us  decoded" $O/rates.log
rm -rf "$O/prof"
timeout -k 10 250 rocprofv3 --kernel-trace -d $O/prof -o v -- python3 $R/tools/misc_rates.py > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 $R/tools/kernel_times.py $O/prof/v_results.db 2>&1 | grep -v "at::native\|rocclr"
