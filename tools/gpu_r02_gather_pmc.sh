#!/bin/bash
# What bounds the dictionary gather with a 512 KB dictionary: L1 <-> L2 requests and L2 hits of the
# tile-per-workgroup kernel (gather through the caches) and of the persistent kernel (first part of
# the dictionary in LDS), one rocprofv3 --pmc pass per counter group.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02e; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="uniform bw16" PQD_RATES_NOCHECK=1
for mode in 0 1; do
  export PQD_DICT_LDS=$mode
  for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout -s KILL 90 rocprofv3 --pmc $grp -d $O/m${mode}_$tag -o c -- python3 $R/tools/sweeps.py dictionary > $O/m${mode}_$tag.log 2>&1 || { tail -3 $O/m${mode}_$tag.log; exit 1; }
    echo "== PQD_DICT_LDS=$mode $grp"
    python3 $R/tools/pmc_summary.py $(find $O/m${mode}_$tag -name "*counter_collection.csv" | head -1) rle_
  done
done
