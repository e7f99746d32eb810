#!/bin/bash
# Round check on the GPU box: parity tests, smoke, the bench line, the rocprofv3 kernel
# summary of the same bench command, and the two HBM traffic counter passes.
# Every GPU step has its own time limit; the script stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r01 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --skip-check > $O/prof_stats.log 2>&1 || { tail -20 $O/prof_stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d $O/pmc_$c -o vm --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-check > $O/pmc_$c.log 2>&1 || { tail -20 $O/pmc_$c.log; exit 1; }
done
find $O/prof_stats -name "*kernel_stats.csv" | head -2
echo round check done
