#!/bin/bash
# cycles the chip counts while the DELTA kernel runs, beside the kernel's duration: the clock it runs at, and how busy the vector units are
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/clock; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cd $R
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $grp | cut -d' ' -f2)
  rm -rf "$O/$tag"
  timeout -s KILL 150 rocprofv3 --pmc $grp -d $O/$tag -o c -- python3 bench.py --no-cpu-baseline --skip-check --steps 10 --warmup 2 > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  python3 tools/pmc_summary.py $(find $O/$tag -name "*_results.db" | head -1) "delta_fast"
done
rm -rf $O/trace
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --skip-check --steps 10 --warmup 2 > $O/trace.log 2>&1 || { tail -3 $O/trace.log; exit 1; }
python3 tools/kernel_times.py $(find $O/trace -name "*_results.db" | head -1) 2>/dev/null | head -8 || true
