#!/usr/bin/env python3
"""Per-kernel durations of ONE bench line at a time, warm-up launches excluded (round-4 verdict, weak item 7: one
`kernel_stats.csv` over the whole default bench mixed two workloads in one row and counted first launches).

    python tools/kstats_by_line.py <out-dir>        (on the GPU box, from the repo root)

For the headline (`bench.py --steps 10 --warmup 2 --no-cpu --no-secondary`) and for every line of the secondary block
(`--only-secondary <line>`) the command runs under `rocprofv3 --kernel-trace` (no counters, no other trace domains), and
<out-dir>/kernel_stats_<line>.csv gets one row per kernel: calls counted, average / min / max / standard deviation in
microseconds over the dispatches that are left when the first DROP dispatches of that kernel are dropped -- the warm-up
steps of the line (the headline's two warm-up launches; one warm call of every secondary line) plus the cold first launch."""
import collections
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = [("headline", ["--steps", "10", "--warmup", "2", "--no-cpu", "--no-secondary"], 2),
         ("lap_any_4gib_ordered", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "lap_any_4gib_ordered"], 1),
         ("lap_any_4gib_init4", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "lap_any_4gib_init4"], 1),
         ("known_lap_79ch_chain_full_payloads", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "known_lap_79ch_chain_full_payloads"], 1),
         ("known_lap_79ch_chain", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "known_lap_79ch_chain"], 1),
         ("clk6_bruteforce", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "clk6_bruteforce"], 1),
         ("clk6_bruteforce_all_types", ["--steps", "2", "--warmup", "1", "--no-cpu", "--only-secondary", "clk6_bruteforce_all_types"], 1)]


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for line, args, drop in LINES:
        d = "/tmp/kstats_%d_%s" % (os.getpid(), line)
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                            sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write("%s: rocprofv3 failed\n%s\n" % (line, r.stderr[-1500:]))
            continue
        per = collections.OrderedDict()
        rows = []
        for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            rows += list(csv.DictReader(open(path)))
        rows.sort(key=lambda x: int(x["Start_Timestamp"]))
        for x in rows:
            per.setdefault(x["Kernel_Name"], []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
        with open(os.path.join(out, "kernel_stats_%s.csv" % line), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsCounted", "CallsDropped", "AverageUs", "MinUs", "MaxUs", "StdDevUs", "TotalUs"])
            for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1][min(drop, len(kv[1]) - 1):])):
                # the kernels of the workload's set-up (synthetic stream, uploads) run once or twice: nothing to drop there
                k = drop if len(v) > drop + 1 else 0
                u = v[k:]
                w.writerow([name, len(u), k, "%.2f" % statistics.mean(u), "%.2f" % min(u), "%.2f" % max(u),
                            "%.2f" % (statistics.pstdev(u) if len(u) > 1 else 0.0), "%.1f" % sum(u)])
        shutil.rmtree(d, ignore_errors=True)
        print(line, "->", "kernel_stats_%s.csv" % line, "(%d kernels)" % len(per))


if __name__ == "__main__":
    main()
