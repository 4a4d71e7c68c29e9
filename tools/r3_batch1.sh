#!/bin/bash
# round-3 batch 1: op rates, correctness of the new sliding-check kernel builds, A/B of all variants
out=gpurun_out/r03_a; mkdir -p $out
tools/valu_rate.bin > $out/valu_rate.txt 2>&1
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/w2t2.so libbtbb_amd/variants/w2t1.so libbtbb_amd/variants/b20.so; do
  echo "== $so" >> $out/tests.txt
  LIBBTBB_AMD_SO=$PWD/$so timeout 900 python -m pytest tests/test_gpu_scan.py -x -q -m gpu 2>&1 | tail -3 >> $out/tests.txt
done
tools/ab_variants.sh 10 > $out/ab1.txt 2>&1
tools/ab_variants.sh 10 > $out/ab2.txt 2>&1
cat $out/tests.txt $out/ab1.txt $out/ab2.txt
