#!/bin/bash
# A/B batch on the GPU box: scan tests for the listed builds, then the headline bench for every variant, twice.
#   tools/ab_batch.sh <out-dir-name> "<so files to test>"
out=gpurun_out/$1; mkdir -p $out
for so in $2; do
  echo "== $so" >> $out/tests.txt
  LIBBTBB_AMD_SO=$PWD/$so timeout 900 python -m pytest tests/test_gpu_scan.py -x -q -m gpu 2>&1 | tail -3 >> $out/tests.txt
done
tools/ab_variants.sh 10 > $out/ab1.txt 2>&1
tools/ab_variants.sh 10 > $out/ab2.txt 2>&1
cat $out/tests.txt; paste $out/ab1.txt $out/ab2.txt | awk '{print $1, $2, $6, $4}'
