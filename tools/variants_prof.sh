#!/bin/bash
# On the GPU box: kernel times of the lines matching $1 of a rates tool ($2..., default tools/misc_rates.py)
# for each library variant in build_exp/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1" PQD_RATES_NOCHECK=1
shift
tool=${1:-tools/misc_rates.py}; shift
for so in $R/build_exp/libpqd_*.so; do
  n=$(basename $so .so)
  export PQD_LIBRARY=$so
  timeout -k 10 120 rocprofv3 --kernel-trace -d $O/var_$n -o v -- python3 $R/$tool "$@" > $O/var_$n.log 2>&1 || { tail -5 $O/var_$n.log; exit 1; }
  echo "== $n"; grep -E "us  decoded" $O/var_$n.log
  python3 $R/tools/kernel_times.py $O/var_$n/v_results.db | grep "pqd::"
done
