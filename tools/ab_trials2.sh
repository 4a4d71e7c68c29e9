#!/bin/bash
# A/B of library builds on the GPU box: packet tests for the listed variants, then the two brute-force bench lines for every build, twice
#   tools/ab_trials2.sh <out-dir-name> "<so files to test>"
out=gpurun_out/$1; mkdir -p $out
for so in $2; do
  echo "== $so" >> $out/tests.txt
  LIBBTBB_AMD_SO=$PWD/$so timeout 1200 python -m pytest tests/test_gpu_packets.py -x -q -m gpu -k "trial or brute or clk or uap" 2>&1 | tail -3 >> $out/tests.txt
done
for rep in 1 2; do
  for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
    for line in clk6_bruteforce clk6_bruteforce_all_types; do
      LIBBTBB_AMD_SO=$PWD/$so timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --only-secondary $line 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d['secondary'].items(): print('$so', k, v['value'], 'kernel_ms', v['roofline']['kernel_ms'], 'parity', v.get('parity'))" >> $out/ab.txt
    done
  done
done
cat $out/tests.txt $out/ab.txt
