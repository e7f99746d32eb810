#!/bin/bash
# kernel trace of the lines of tools/misc_rates.py whose name contains $1
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/ratetrace; mkdir -p $O; rm -rf $O/t
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/t -o w -- python3 $R/tools/misc_rates.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
grep "$1" $O/trace.log || true
python3 $R/tools/kernel_times.py $(find $O/t -name "*_results.db" | head -1) | grep -v "rocclr\|at::native"
