#!/bin/bash
# A/B of library builds on the GPU box through bench.py's secondary block (config 3 chain, config 5 brute force)
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  LIBBTBB_AMD_SO=$PWD/$so timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$so', ' | '.join('%s %s %s ms=%s kernel_ms=%s' % (k, v['value'], v['unit'], v['ms_per_step'], v['roofline']['kernel_ms']) for k,v in d['secondary'].items()))"
done
