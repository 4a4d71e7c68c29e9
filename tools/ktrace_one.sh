#!/bin/bash
# timeline of the last step of ONE bench invocation under rocprofv3 --kernel-trace: tools/ktrace_one.sh <out-file> <n-last-kernels> <bench args...>
out=$1; last=$2; shift 2
export TMPDIR=/tmp
d=/tmp/ktrace_one_$$
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py "$@" > /dev/null 2>&1 )
python - "$d" "$last" > "$out" <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for x in rows:
    print("%9.1f .. %9.1f us  (%8.1f)  q%-3s %s" % ((int(x["Start_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - t0) / 1e3,
          (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3, x.get("Queue_Id", "?"), x["Kernel_Name"].split("(")[0][:60]))
PY
rm -rf $d
