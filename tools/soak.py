#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak for the scan path (run on the MI355X box; not part of pytest):
random stream lengths, strides, error cycles, LAP_ANY / known LAP, max_ac_errors, btbb_init sizes,
search windows that end inside words, multi-stream launches with small hit buffers.  Prints one
JSON line with the number of comparisons; exits non-zero on the first mismatch.

  python tools/soak.py --seconds 120 --seed 1"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import libbtbb_amd as bt
from libbtbb_amd import synth
import _libs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    orc = _libs.oracle()
    lib = bt.lib()
    t_end = time.time() + a.seconds
    done = {"cases": 0, "hits": 0, "by_init": {}}
    inits = (2, 3, 1, 4, 5)
    for n_init in inits:
        lib.btbbx_shutdown()
        bt.init(n_init)
        orc.orc_reset_syndrome_map()
        orc.orc_init(n_init)
        t_part = time.time() + a.seconds / len(inits)
        while time.time() < min(t_end, t_part):
            # mostly short streams; one in ten spans several hundred tiles (XCD-partitioned tile order)
            nwords = int(rng.integers(1 << 18, 1 << 20)) if rng.random() < 0.1 else int(rng.integers(2, 1 << 13))
            stride = int(rng.choice([512, 600, 1024, 4096]))
            known = rng.random() < 0.35
            lap = int(rng.integers(0, 1 << 24)) if known else None
            words, _ = synth.make_stream(int(rng.integers(0, 1 << 60)), nwords, stride=stride, lap=lap,
                                         err_cycle=int(rng.integers(1, 7)))
            if rng.random() < 0.2:          # dense: many sync words back to back
                sw = synth.syncword(int(rng.integers(0, 1 << 24)))
                words[: min(nwords, 64)] = np.uint64(sw)
            sym = np.ascontiguousarray(synth.unpack_bits(words))
            search = int(rng.integers(1, nwords * 64 - 63 + 1))
            me = int(rng.integers(0, 6))
            got = bt.scan_words(words, search, lap=bt.LAP_ANY if lap is None else lap, max_ac_errors=me)
            want = _libs.orc_find_all(sym, search, _libs.LAP_ANY if lap is None else lap, me)
            g = [(int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in got]
            if g != want:
                print(json.dumps({"mismatch": dict(n_init=n_init, nwords=nwords, stride=stride, lap=lap, search=search, me=me,
                                                   got=len(g), want=len(want))}))
                return 1
            done["cases"] += 1
            done["hits"] += len(g)
            done["by_init"][n_init] = done["by_init"].get(n_init, 0) + 1
    print(json.dumps(done))
    return 0


if __name__ == "__main__":
    sys.exit(main())
