#!/bin/bash
# RLE / dictionary parity tests, then the bench with the RLE_DICTIONARY extra
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rle.py -x -q --timeout 120 --timeout-method thread > gpurun_out/tr.log 2>&1 || { tail -30 gpurun_out/tr.log; exit 1; }
tail -1 gpurun_out/tr.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/br.log 2>&1 || { tail -20 gpurun_out/br.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/br.log"):
    if l.startswith("{"):
        d = json.loads(l)
        e = d["extra"]["rle_dictionary_int64"]
        print("delta kernel_ms", d["roofline"]["kernel_ms"], "| rle-dict kernels_ms", e["kernels_ms"], "decoded GB/s", e["decoded_GBps"], "frac", e["roofline"]["frac"])
PY
