#!/bin/bash
# DELTA_BINARY_PACKED int64, 2048 pages of 128 Ki values, one miniblock width per run:
# kernel trace, the two HBM traffic counter passes and the instruction mix, each its own run.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02dw; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PQD_SWEEP_DELTA_REPLICATE=512
for w in "(width 0)" "width 8" "width 12" "width 16" "width 32" "(width 64)"; do
  export PQD_RATES_ONLY="$w"
  tag=$(echo "$w" | tr -dc '0-9')
  echo "== miniblock width $tag"
  rm -rf "$O/t$tag"
  timeout -s KILL 150 rocprofv3 --kernel-trace -d $O/t$tag -o v -- python3 $R/tools/sweeps.py delta > $O/t$tag.log 2>&1 || { tail -5 $O/t$tag.log; exit 1; }
  grep "us  decoded" $O/t$tag.log
  python3 $R/tools/kernel_times.py $O/t$tag/v_results.db 2>&1 | grep "delta\|kernel  "
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    c=$(echo $grp | cut -d' ' -f1)
    rm -rf "$O/p$tag$c"
    timeout -s KILL 150 rocprofv3 --pmc $grp -d $O/p$tag$c -o c -- python3 $R/tools/sweeps.py delta > $O/p$tag$c.log 2>&1 || { tail -5 $O/p$tag$c.log; exit 1; }
    python3 $R/tools/pmc_summary.py $(find $O/p$tag$c -name "*_results.db" | head -1) delta_fast
  done
done
rm -rf "$O"
echo delta widths done
