#!/bin/bash
# phase profile of the LAP_ANY kernel: builds made with -DSCAN_PROFILE print a table per launch
for so in libbtbb_amd/variants/p*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  LIBBTBB_AMD_SO=$PWD/$so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-secondary 2>&1 | grep "scan profile" | tail -2
done
