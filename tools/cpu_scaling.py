#!/usr/bin/env python3
"""How the reference's all-matches scan scales over the host threads of this box (diagnostic for bench.py's
cpu_baseline): cgroup CPU quota, then the native multi-thread timing (oracle/ref_internals.c refint_find_all_mt) at
1 .. all threads, pinned to distinct physical cores first.  CPU only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _libs  # noqa: E402
import bench  # noqa: E402
from libbtbb_amd import synth  # noqa: E402

out = {}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        out[f] = open(f).read().strip()
    except OSError:
        pass
allowed, firsts = bench.physical_cpus()
out["allowed"] = len(allowed)
out["physical"] = len(firsts)
nw = 1 << 24                                             # 2^30 symbols
words = synth.make_stream(bench.SEED, nw, stride=4096)
if isinstance(words, tuple):
    words = words[0]
ref = _libs.ref()
ref.btbb_init(2)
sym = _libs.ref_unpack_mt(words, 32)
n = len(sym) - 63
rows = []
order = firsts + [c for c in allowed if c not in firsts]
t = 1
while True:
    k = min(t, len(order))
    n_k = min(n, k << 24)                                # 16 M symbols per thread
    off, laps, errs, found, secs, wall = _libs.ref_find_all_mt(sym, n_k, 0xFFFFFFFF, 2, k, order[:k])
    rows.append({"threads": k, "Gbit_s": round(n_k / wall / 1e9, 3), "per_thread_Msym_s": round(float((n_k / k / secs / 1e6).mean()), 1),
                 "wall_s": round(wall, 3)})
    if k == len(order):
        break
    t *= 2
out["scaling"] = rows
# the same with all threads on noise only (no hits): is it the hit path?
print(json.dumps(out, indent=1))
