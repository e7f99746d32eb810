#!/bin/bash
# the wide table: kernel times of an 8-row-group share under rocprofv3, then the whole 64 x 64 table on one GPU
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02g; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/wide8 -o w -- python3 $R/bench.py --config wide --wide-row-groups 8 --steps 3 --warmup 1 > $O/wide8.json 2> $O/wide8.err || { tail -5 $O/wide8.err; exit 1; }
python3 $R/tools/kernel_times.py $O/wide8/w_results.db | grep -v "at::\|rocclr\|elementwise" > $O/wide8_kernels.txt; cat $O/wide8_kernels.txt
cd $R
timeout -k 10 900 python3 -u bench.py --config wide --steps 5 > $O/wide64.json 2> $O/wide64.err; echo "wide64 rc=$?"; cat $O/wide64.json; tail -3 $O/wide64.err
