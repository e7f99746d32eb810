#!/bin/bash
# counters of the kernels of the default bench step: bash tools/gpu_pmc_bench.sh <kernel match> "<counters>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmcb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
match=$1; shift
for grp in "$@"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf "$O/$tag"
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $O/$tag -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-check > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $(find $O/$tag -name "*counter_collection.csv" -o -name "*_results.db" | head -1) $match
done
rm -rf "$O"
