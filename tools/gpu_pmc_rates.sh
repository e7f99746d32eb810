#!/bin/bash
# counters ($3...) of the kernels matching $2 while tools/misc_rates.py runs the lines matching $1
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmcrates; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1"; match="$2"; shift 2
rm -rf "$O/run"
timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/run -o c -- python3 $R/tools/misc_rates.py > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
python3 $R/tools/pmc_summary.py $(find $O/run -name "*_results.db" | head -1) "$match"
