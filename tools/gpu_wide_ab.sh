#!/bin/bash
# the wide-table bench with several builds of the library, three rounds
set -e
O=gpurun_out/vpt; mkdir -p $O
for round in 1 2 3; do
for v in "$@"; do
  PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 400 python bench.py --config wide --no-cpu-baseline --steps 8 --warmup 2 > $O/wide_$v.json 2> $O/wide_$v.err || { tail -5 $O/wide_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/wide_$v.json"))
print("wide $v", d["value"], d["ms_per_step"])
PY
done
done
