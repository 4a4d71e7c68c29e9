#!/bin/bash
# per-kernel durations of ONE bench invocation under rocprofv3 --kernel-trace (warm-ups included): tools/kstats_one.sh <out-file> <bench args...>
out=$1; shift
export TMPDIR=/tmp
d=/tmp/kstats_one_$$
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py "$@" > /dev/null 2>&1 )
python - "$d" > "$out" <<'PY'
import csv, glob, sys, collections, statistics
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
per = collections.OrderedDict()
for x in sorted(rows, key=lambda x: int(x["Start_Timestamp"])):
    per.setdefault(x["Kernel_Name"].split("(")[0], []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    u = v[1:] if len(v) > 2 else v
    big = [x for x in u if x > 0.1 * max(u)]              # (a gated kernel returns at once most of the time: its real runs apart)
    print("%-70s calls %3d  avg %9.2f  min %9.2f  max %9.2f us" % (name[:70], len(v), statistics.mean(u), min(u), max(u))
          + ("   runs > max/10: %d avg %9.2f median %9.2f" % (len(big), statistics.mean(big), statistics.median(big)) if len(big) != len(u) else "   median %9.2f" % statistics.median(u)))
PY
rm -rf $d
