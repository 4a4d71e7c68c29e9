#!/bin/bash
# A/B on the GPU box: LAP_ANY scan rate with tables for the listed error counts (tools/init_sweep.py) for the normal build and every variant
for rep in 1 2; do
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  echo "$so $(LIBBTBB_AMD_SO=$PWD/$so SWEEP_N=${1:-4} timeout 300 python tools/init_sweep.py 2>/dev/null | tail -1)"
done
done
