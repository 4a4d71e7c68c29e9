#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench (no CPU baselines): per-kernel average durations
#   tools/kstats.sh <out-dir-name> [pattern]
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1 )
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
python - <<PY
import csv
for r in csv.DictReader(open("$out/kernel_stats.csv")):
    if "${2:-}" in r["Name"] and float(r["TotalDurationNs"]) > 2e4:
        print("%-64s calls %4s avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
