#!/bin/bash
# tables for four errors (scan_slide_kernel<Slide4>): time and fabric traffic of the normal build and every variant
#   tools/ab_init4_traffic.sh <out-dir-name>
out=gpurun_out/$1; mkdir -p $out
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  SWEEP_N=3,4 LIBBTBB_AMD_SO=$PWD/$so timeout 300 python tools/init_sweep.py 2>/dev/null | tail -1 > $out/sweep_$n.json
  SWEEP_N=4 LIBBTBB_AMD_SO=$PWD/$so timeout 300 python tools/pmc_collect.py --out $out/pmc_$n --kernel scan_ --groups FETCH_SIZE WRITE_SIZE TCC_HIT_sum,TCC_MISS_sum -- python tools/init_sweep.py > $out/pmc_$n.json 2> $out/pmc_$n.err
  rm -rf $out/pmc_$n
  echo "== $so"; cat $out/sweep_$n.json; python -c "
import json; d=json.load(open('$out/pmc_$n.json'))
for k,v in d.items(): print(k, {c: round(x['mean_per_launch']) for c,x in v.items()})"
done
