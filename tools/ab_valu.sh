#!/bin/bash
# per-variant instruction counts of one kernel: tools/ab_valu.sh <kernel-substring>
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  LIBBTBB_AMD_SO=$PWD/$so python tools/pmc_collect.py --out /tmp/abv_$(basename $so .so) --kernel ${1:-decode_hits} --groups SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES -- python bench.py --steps 1 --warmup 1 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    w=v['SQ_WAVES']['mean_per_launch']
    print('$so', k[:30], 'per wave: VALU %.0f SALU %.0f LDS %.0f' % (v['SQ_INSTS_VALU']['mean_per_launch']/w, v['SQ_INSTS_SALU']['mean_per_launch']/w, v['SQ_INSTS_LDS']['mean_per_launch']/w))"
done
