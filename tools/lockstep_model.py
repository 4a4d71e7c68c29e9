#!/usr/bin/env python3
"""How many lock-step passes the LAP_ANY survivor loop needs, for the arrangement the kernel uses and for the
alternatives NOTEBOOK.md 6.3 / 9 talk about -- from the real barker pre-filter over a random stream, not from a binomial
guess (windows at neighbouring offsets exclude each other, so counts per 32 offsets are narrower than binomial).
CPU only:  python tools/lockstep_model.py [log2 of the number of symbols, default 24]

A wave owns 64 consecutive 64-bit words of each of its tiles (one word per lane and tile); a trip works on UNROLL
tiles; a "chain" is one 32-offset half of a word and a pass takes one survivor of every chain of every lane, so a
trip costs max-over-lanes-and-chains(count) passes of 2 * UNROLL chain slots each."""
import sys

import numpy as np

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
rng = np.random.default_rng(7)
bits = rng.integers(0, 2, n + 64, dtype=np.uint8)
# 7-bit window of offset o = stream bits o+57 .. o+63, bit k of the window = stream[o + 57 + k]
win = np.zeros(n, dtype=np.uint8)
for k in range(7):
    win |= bits[57 + k:57 + k + n] << k
dist = lambda v, c: np.array([bin(x ^ c).count("1") for x in range(128)], dtype=np.uint8)[v]
surv = (dist(win, 0x27) <= 1) | (dist(win, 0x58) <= 1)                  # BARKER_DISTANCE <= 1
print("survivors: %.4f of the offsets" % surv.mean())

halves = surv.reshape(-1, 32).sum(axis=1)                               # per 32-offset half
print("per half: mean %.2f  sd %.2f  (binomial(32, 1/8) would have sd %.2f)" % (halves.mean(), halves.std(), (32 * 7 / 64) ** 0.5))
words = halves.reshape(-1, 2)                                           # [word][half]
n_waves = words.shape[0] // 64
lane = words[:n_waves * 64].reshape(n_waves, 64, 2)                     # [wave-tile][lane][half]


def report(name, passes, chains, offsets_per_lane):
    slots = passes.mean() * chains
    useful = lane.sum() / (n_waves * 64) * (offsets_per_lane / 64)
    print("%-46s passes/trip %5.2f  chain slots per 64 offsets %5.2f  density %.0f %%"
          % (name, passes.mean(), slots * 64 / offsets_per_lane, 100 * useful / slots))


for unroll in (1, 2, 3, 4):
    t = lane[:(n_waves // unroll) * unroll].reshape(-1, unroll, 64, 2)
    report("halves as chains, %d tile(s) per trip" % unroll, t.max(axis=(1, 2, 3)), 2 * unroll, 64 * unroll)
t = lane[:(n_waves // 2) * 2].reshape(-1, 2, 64, 2)
report("whole words as chains, 2 tiles per trip", t.sum(axis=3).max(axis=(1, 2)), 2, 128)
report("one chain per lane over both words", t.sum(axis=(1, 3)).max(axis=1), 1, 128)
report("4 chains per lane, perfectly balanced", np.ceil(t.sum(axis=(1, 3)) / 4).max(axis=1), 4, 128)
report("4 chains per lane, balanced over the wave", np.ceil(t.sum(axis=(1, 2, 3)) / 256), 4, 128)
