#!/bin/bash
# A/B of library builds: known-LAP scan of 4 GiB (tools/known_lap_time.py) for the normal build and every variant
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  echo "$so $(LIBBTBB_AMD_SO=$PWD/$so timeout 100 python tools/known_lap_time.py 2>/dev/null | tr '\n' ' ')"
done
