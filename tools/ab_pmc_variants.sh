#!/bin/bash
# instruction counts of the headline kernel for the normal build and every variant: tools/ab_pmc_variants.sh <out-dir-name>
out=gpurun_out/$1; mkdir -p $out
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  LIBBTBB_AMD_SO=$PWD/$so timeout 300 python tools/pmc_collect.py --out $out/pmc_$n --kernel scan_slide --groups SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY \
    -- python bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $out/pmc_$n.json 2> $out/pmc_$n.err
  rm -rf $out/pmc_$n
  echo "== $n $(LIBBTBB_AMD_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_median"], d["config"]["hits_per_gpu"])')"
  python -c "
import json; d=json.load(open('$out/pmc_$n.json'))
for k,v in d.items(): print(' ', k, {c: round(x['mean_per_launch']/1e6,1) for c,x in v.items()})"
done
