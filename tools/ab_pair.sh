#!/bin/bash
# A/B of two library builds on the GPU box, alternating order, n rounds: tools/ab_pair.sh <a.so> <b.so> [rounds] [steps]
a=$1; b=$2; n=${3:-5}; steps=${4:-10}
one() { LIBBTBB_AMD_SO=$PWD/$1 timeout 300 python bench.py --steps $steps --warmup 2 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel_ms"], r["kernel_ms_median"], r["kernel_ms_min"], d["config"]["hits_per_gpu"])'; }
for i in $(seq $n); do
  if [ $((i % 2)) = 1 ]; then echo "A $a $(one $a)"; echo "B $b $(one $b)"; else echo "B $b $(one $b)"; echo "A $a $(one $a)"; fi
done
