#!/bin/bash
# RLE_DICTIONARY: parity of the three gather paths, then the width sweep with the LDS path off and on.
set -o pipefail
out=gpurun_out/r02b
mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rle_lds.py tests/test_gpu_rle.py -x -q --timeout 120 --timeout-method thread > $out/tests.txt 2>&1
rc=$?; tail -15 $out/tests.txt; [ $rc -ne 0 ] && exit $rc
PQD_DICT_LDS=0 timeout -k 10 200 python -u tools/sweeps.py dictionary > $out/sweep_cache.txt 2>&1 || exit 1
PQD_DICT_LDS=2 PQD_DICT_LDS_MAX_BYTES=100000000 timeout -k 10 200 python -u tools/sweeps.py dictionary > $out/sweep_lds.txt 2>&1 || exit 1
echo "--- cache"; cat $out/sweep_cache.txt; echo "--- lds"; cat $out/sweep_lds.txt
