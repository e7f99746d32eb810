#!/bin/bash
# when the workgroups of delta_fast_kernel end, with the priority left alone and with it going round
# (build_exp/libpqd_t_rot0.so, libpqd_t_rot4.so: tools/build_variant.sh t_rot0 -DPQD_DELTA_TIMING -DPQD_DELTA_ROTATE=0; ... t_rot4 -DPQD_DELTA_TIMING)
set -e
O=gpurun_out/lifetimes; mkdir -p $O
for v in t_rot0 t_rot4; do
  echo "== $v" >> $O/out.txt
  PQD_LIBRARY=build_exp/libpqd_$v.so timeout -k 10 200 python tools/delta_timing.py 2048 2>/dev/null | grep -v "^  xcc\|^trial [012]" >> $O/out.txt
done
cat $O/out.txt
