#!/bin/bash
# On the GPU box: the calibration kernels of tools/fetch_calib.hip under rocprofv3 counter passes (never combined with traces
# other than --kernel-trace) -> gpurun_out/<dir>/fetch_calib.txt: per dispatch, in launch order, the kernel, its duration and
# the counters.   tools/fetch_calib_run.sh r05_calib
out=gpurun_out/${1:-r05_calib}; mkdir -p $out; export TMPDIR=/tmp
for grp in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  ( cd /tmp && rm -rf /tmp/fc_$$ && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/fc_$$ -- $GRAFT_REPO_ROOT/tools/fetch_calib.bin > /dev/null 2>&1 )
  python - "$tag" /tmp/fc_$$ >> $out/fetch_calib.txt <<'PY'
import csv, glob, sys, collections
tag, d = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        key = int(r["Dispatch_Id"])
        rows.setdefault(key, {"kernel": r["Kernel_Name"].split("(")[0]})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("== pass", tag)
for k in sorted(rows):
    v = rows[k]
    print(k, v["kernel"], "ms=%.3f" % dur.get(k, -1), " ".join("%s=%.1f" % (c, x) for c, x in v.items() if c != "kernel"))
PY
  rm -rf /tmp/fc_$$
done
cat $out/fetch_calib.txt
