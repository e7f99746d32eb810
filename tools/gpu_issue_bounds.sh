#!/bin/bash
# scalar- and vector-issue time of every kernel of a tool script beside its duration:
# bash tools/gpu_issue_bounds.sh tools/misc_rates.py
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/issue; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
script=$R/$1; shift
export PQD_RATES_NOCHECK=1
timeout -s KILL 250 rocprofv3 --kernel-trace -d /tmp/ib_t -o t -- python3 $script "$@" > $O/trace.log 2>&1 || { tail -3 $O/trace.log; exit 1; }
timeout -s KILL 250 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES -d /tmp/ib_p -o c -- python3 $script "$@" > $O/pmc.log 2>&1 || { tail -3 $O/pmc.log; exit 1; }
python3 $R/tools/issue_bounds.py $(find /tmp/ib_p -name "*counter_collection.csv" -o -name "*_results.db" | head -1) /tmp/ib_t/t_results.db
