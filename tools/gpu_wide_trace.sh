#!/bin/bash
# kernel trace of the wide-table bench: which kernels its step is made of
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/widetrace; mkdir -p $O; rm -rf $O/t
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/t -o w -- python3 $R/bench.py --config wide --no-cpu-baseline --skip-check --steps 4 --warmup 1 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 $R/tools/kernel_times.py $(find $O/t -name "*_results.db" | head -1) > $O/kernels.txt
cat $O/kernels.txt
tail -1 $O/trace.log | cut -c1-300
