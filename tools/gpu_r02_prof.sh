#!/bin/bash
# kernel times (rocprofv3 --kernel-trace) of the sweep lines matching $1, dictionary-in-LDS path off and on
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02c
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1" PQD_DICT_LDS_MAX_BYTES=100000000
for mode in 0 2; do
  export PQD_DICT_LDS=$mode
  timeout -k 10 150 rocprofv3 --kernel-trace -d $O/lds$mode -o v -- python3 $R/tools/sweeps.py dictionary > $O/lds$mode.log 2>&1 || { tail -5 $O/lds$mode.log; exit 1; }
  echo "== PQD_DICT_LDS=$mode"; grep -E "us  decoded" $O/lds$mode.log
  python3 $R/tools/kernel_times.py $O/lds$mode/v_results.db | grep -E "rle_|kernel  " 
done
