#!/bin/bash
# decode_hits_kernel by packet type for the normal build and every library under libbtbb_amd/variants (GPU box):
#   tools/ab_decode_time.sh OUTFILE [decode_time.py cases ...]
out=$1; shift
: > $out
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  echo "== $so" >> $out
  LIBBTBB_AMD_SO=$PWD/$so timeout 300 python tools/decode_time.py "$@" >> $out 2>/dev/null
done
