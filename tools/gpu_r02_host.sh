#!/bin/bash
# Host entry points on pooled workspaces: tests, per-call rates, differential soak.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/host_tests.log 2>&1 || { tail -30 gpurun_out/host_tests.log; exit 1; }
tail -3 gpurun_out/host_tests.log
timeout -k 10 300 python -u tools/host_rate.py > gpurun_out/host_rate.log 2>&1 || { tail -30 gpurun_out/host_rate.log; exit 1; }
cat gpurun_out/host_rate.log
PQD_SOAK_HOST=1 PQD_SOAK_SECONDS=60 timeout -k 10 200 python -u tools/soak.py 7 > gpurun_out/host_soak.log 2>&1 || { tail -30 gpurun_out/host_soak.log; exit 1; }
tail -3 gpurun_out/host_soak.log
