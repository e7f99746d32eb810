#!/bin/bash
# instruction mix of the kernels matching $2 while tools/misc_rates.py runs the lines matching $1
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="$1"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f2)
  rm -rf "$O/$tag"
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $O/$tag -o c -- python3 $R/tools/misc_rates.py > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  python3 $R/tools/pmc_summary.py $(find $O/$tag -name "*_results.db" | head -1) "$2"
done
