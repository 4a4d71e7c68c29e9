#!/bin/bash
# A/B on the GPU box: the two config-3 chain lines of the bench's secondary block + the 4 GiB known-LAP scan (tools/measure_paths.py is
# too broad for this) for the normal build and every variant, twice
for rep in 1 2; do
  for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
    [ -f "$so" ] || continue
    for line in known_lap_79ch_chain known_lap_79ch_chain_full_payloads; do
      LIBBTBB_AMD_SO=$PWD/$so timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --only-secondary $line 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d['secondary'].items(): print('$so', k, 'ms', v['ms_per_step'], 'scan kernel_ms', v['roofline'].get('kernel_ms'), 'parity', v.get('parity'))"
    done
  done
done
