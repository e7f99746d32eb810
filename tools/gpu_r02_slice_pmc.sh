#!/bin/bash
# The sliced dictionary kernel at a 512 KB dictionary (index width 16): one rocprofv3 --pmc pass per
# counter group given on the command line (each argument one group), or a default set.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="${PQD_RATES_ONLY:-uniform bw16}" PQD_RATES_NOCHECK=1
if [ $# -eq 0 ]; then
  set -- "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
fi
for grp in "$@"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf "$O/$tag"
  timeout -s KILL 90 rocprofv3 --pmc $grp -d $O/$tag -o c -- python3 $R/tools/sweeps.py dictionary > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $(find $O/$tag -name "*counter_collection.csv" -o -name "*_results.db" | head -1) rle_
done
rm -rf "$O"
