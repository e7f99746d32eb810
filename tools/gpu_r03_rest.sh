#!/bin/bash
# beside the bench line: encoder rates (and their kernels), the sweeps, the wide table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r03r; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 -u "$R/tools/encode_rates.py" > "$O/encode_rates.txt" 2> "$O/encode_rates.err" || { tail -20 "$O/encode_rates.err"; exit 1; }
cat "$O/encode_rates.txt"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$O/enc_prof" -o v -- python3 "$R/tools/encode_rates.py" > "$O/enc_prof.log" 2>&1 || { tail -5 "$O/enc_prof.log"; exit 1; }
python3 "$R/tools/kernel_times.py" "$O/enc_prof/v_results.db" > "$O/encode_kernels.txt" 2>&1
timeout -k 10 400 python3 -u "$R/tools/sweeps.py" > "$O/sweeps.log" 2>&1 || { tail -20 "$O/sweeps.log"; exit 1; }
grep "us  decoded" "$O/sweeps.log" > "$O/sweeps.txt"
cd "$R"
timeout -k 10 300 python3 -u bench.py --config wide --no-cpu-baseline > "$O/wide.json" 2> "$O/wide.err" || { tail -20 "$O/wide.err"; exit 1; }
cat "$O/wide.json"
