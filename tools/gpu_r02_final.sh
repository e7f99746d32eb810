#!/bin/bash
# the measurement set beside the bench line: other encodings (rates + kernels), the sweeps, the wide table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
bash $R/tools/gpu_r02_misc.sh > $O/misc.log 2>&1 || { tail -20 $O/misc.log; exit 1; }
timeout -k 10 400 python3 -u $R/tools/sweeps.py > $O/sweeps.log 2>&1 || { tail -20 $O/sweeps.log; exit 1; }
grep "us  decoded" $O/sweeps.log | tail -60
cd $R
timeout -k 10 300 python3 -u bench.py --config wide --no-cpu-baseline > $O/wide.json 2> $O/wide.err || { tail -20 $O/wide.err; exit 1; }
cat $O/wide.json
