#!/bin/bash
# counters of the kernels behind one line of tools/misc_rates.py: bash tools/gpu_r02_pmc_misc.sh "<line match>" <kernel match> "<counters>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02pm; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1" PQD_RATES_NOCHECK=1
match=$2; shift 2
for grp in "$@"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf "$O/$tag"
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $O/$tag -o c -- python3 $R/tools/misc_rates.py > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $(find $O/$tag -name "*counter_collection.csv" -o -name "*_results.db" | head -1) $match
done
rm -rf "$O"
