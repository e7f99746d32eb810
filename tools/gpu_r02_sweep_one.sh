#!/bin/bash
# kernels of the lines of tools/sweeps.py <$1> matching $2 under rocprofv3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02sw
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$2"
rm -rf "$O/prof"
timeout -k 10 250 rocprofv3 --kernel-trace -d $O/prof -o v -- python3 $R/tools/sweeps.py $1 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
grep "us  decoded" $O/prof.log
python3 $R/tools/kernel_times.py $O/prof/v_results.db 2>&1 | grep -v "at::native\|rocclr\|elementwise"
