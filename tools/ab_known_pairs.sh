#!/bin/bash
# alternating A/B of two builds on the 4 GiB known-LAP scan (tools/known_lap_time.py): tools/ab_known_pairs.sh <a.so> <b.so> [rounds]
a=$1; b=$2; n=${3:-4}
one() { LIBBTBB_AMD_SO=$PWD/$1 python tools/known_lap_time.py 2>/dev/null | tr '\n' ' '; }
for i in $(seq $n); do
  if [ $((i % 2)) = 1 ]; then echo "A $(one $a)"; echo "B $(one $b)"; else echo "B $(one $b)"; echo "A $(one $a)"; fi
done
