#!/usr/bin/env python3
"""Run a command under `rocprofv3 --pmc <counters>` (one pass per counter group, never combined
with sys/runtime traces) and print per-kernel means as JSON.

  python tools/pmc_collect.py --out gpurun_out/pmc --groups FETCH_SIZE WRITE_SIZE -- python bench.py --steps 4 --warmup 1 --no-cpu

Each --groups item is one pass; counters inside an item are comma separated."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--groups", nargs="+", required=True)
    ap.add_argument("--kernel", default="", help="substring filter on kernel names")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    env = dict(os.environ, TMPDIR="/tmp")
    result = collections.defaultdict(dict)
    for gi, group in enumerate(a.groups):
        d = os.path.join(a.out, "pass%d" % gi)
        os.makedirs(d, exist_ok=True)
        r = subprocess.run(["rocprofv3", "--pmc"] + group.split(",") + ["--output-format", "csv", "-d", d, "--"] + cmd,
                           env=env, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
            return r.returncode
        vals = collections.defaultdict(list)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].split("(")[0]
                if a.kernel in name:
                    vals[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (name, ctr), v in vals.items():
            result[name][ctr] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    print(json.dumps(result, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
