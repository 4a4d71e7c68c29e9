#!/bin/bash
# A/B of library builds on the GPU box: rocprofv3 kernel averages of the secondary block's kernels matching a pattern
#   tools/ab_decode.sh [pattern]
export TMPDIR=/tmp
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/*.so; do
  [ -f "$so" ] || continue
  d=/tmp/abd_$(basename $so .so); rm -rf $d
  ( cd /tmp && LIBBTBB_AMD_SO=$GRAFT_REPO_ROOT/$so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu > /dev/null 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "${1:-decode_hits}" in r["Name"]:
        print("%-40s %-40s calls %4s avg %9.1f us" % ("$so", r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
