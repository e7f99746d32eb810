#!/bin/bash
# Builds of the library side by side (build_exp/libpqd_NAME.so, tools/build_variant.sh): the DELTA parity tests on each, then the bench line twice, the width sweep (PQD_VPT_SWEEP=1) and the wide table (PQD_VPT_WIDE=1)
set -e
O=gpurun_out/vpt; mkdir -p $O
for v in "$@"; do
  PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 400 python -u -m pytest tests/test_gpu_delta.py -x -q --timeout 120 --timeout-method thread > $O/tests_$v.log 2>&1 || { tail -30 $O/tests_$v.log; exit 1; }
  tail -1 $O/tests_$v.log
done
for v in "$@" "$@"; do
  PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print("$v", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
if [ -n "$PQD_VPT_SWEEP" ]; then
for v in "$@"; do
  PQD_SWEEP_DELTA_REPLICATE=512 PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 300 python tools/sweeps.py delta > $O/sweep_$v.txt 2>&1 || { tail -5 $O/sweep_$v.txt; exit 1; }
  echo "== $v"; grep -i "DELTA" $O/sweep_$v.txt
done
fi
if [ -n "$PQD_VPT_WIDE" ]; then
for v in "$@"; do
  PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 400 python bench.py --config wide --no-cpu-baseline --steps 5 --warmup 2 > $O/wide_$v.json 2> $O/wide_$v.err || { tail -5 $O/wide_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/wide_$v.json"))
print("wide $v", d["value"], d["ms_per_step"])
PY
done
fi
