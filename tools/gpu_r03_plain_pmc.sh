#!/bin/bash
# PLAIN byte arrays through the one-pass tile kernel: kernel durations and the HBM traffic counters of the 32 Ki line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r03p; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
export PQD_RATES_ONLY="PLAIN byte arrays urls (32 Ki)"
timeout -k 10 200 rocprofv3 --kernel-trace -d "$O/trace" -o v -- python3 "$R/tools/misc_rates.py" > "$O/trace.log" 2>&1 || { tail -5 "$O/trace.log"; exit 1; }
python3 "$R/tools/kernel_times.py" "$O/trace/v_results.db" | grep -i "plain\|kernel " > "$O/kernels.txt"
cat "$O/kernels.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$O/pmc_$c" -o vm --output-format csv -- python3 "$R/tools/misc_rates.py" > "$O/pmc_$c.log" 2>&1 || { tail -5 "$O/pmc_$c.log"; exit 1; }
  python3 "$R/tools/pmc_summary.py" "$O/pmc_$c/vm_counter_collection.csv" plain_bytes | tee "$O/pmc_$c.txt"
done
