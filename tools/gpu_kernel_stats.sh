#!/bin/bash
# Per-kernel average durations of one tool script: bash tools/gpu_kernel_stats.sh tools/levels_rates.py [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/kstats
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
script=$R/$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kstats -o ks -- python3 $script "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/tools/kernel_times.py /tmp/kstats/ks_results.db
