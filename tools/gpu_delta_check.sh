#!/bin/bash
# delta parity tests, then the bench at three page counts; stops at the first failure
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_delta.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
: > gpurun_out/b.log
for p in 256 2048 4096; do
  timeout -k 10 120 python bench.py --pages $p --no-cpu-baseline --no-extra >> gpurun_out/b.log 2>&1 || { tail -20 gpurun_out/b.log; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/b.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"]["workload"].split(",")[1].strip(), "value", d["value"], "ms", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
PY
