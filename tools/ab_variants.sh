#!/bin/bash
# A/B of kernel builds on the GPU box: every libbtbb_amd/variants/*.so through the headline bench.
# usage (from the repo root): tools/ab_variants.sh [steps]
steps=${1:-10}
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  out=$(LIBBTBB_AMD_SO=$PWD/$so timeout 300 python bench.py --steps $steps --warmup 2 --no-cpu --no-secondary 2>/dev/null | tail -1)
  echo "$so $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["hits_per_gpu"])' 2>/dev/null)"
done
