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

# ---- round 6: the two arrangements the round-5 verdict asked to price (its item 1a), with what they cost per pass ----
# (a) two chains per 64-offset word walking towards each other: the bottom one takes ceil(n / 2) survivors, the top one the rest
# (b) its cousin: two upward chains per word, split at the multiple of 4 that balances the halves best
# Passes per trip = max over the wave's lanes and words.  What the kernel runs is six fixed passes + a tail, so the figure that
# compares is max(6, passes).
words64 = surv.reshape(-1, 64)
nw = words64.sum(axis=1)
two_ended = np.ceil(nw / 2).astype(np.int64)
cum = np.cumsum(words64, axis=1)
cuts = np.arange(4, 64, 4)
lo = cum[:, cuts - 1]                                                   # survivors below each candidate cut
worst = np.maximum(lo, nw[:, None] - lo)
cousin = worst.min(axis=1)
for name, per_word in (("two chains per word from both ends", two_ended), ("two upward chains, cut at the best multiple of 4", cousin)):
    w = per_word[:(len(per_word) // 128) * 128].reshape(-1, 128)       # 64 lanes x 2 tiles
    passes = w.max(axis=1)
    fixed6 = np.maximum(passes, 6)
    print("%-52s passes/trip %5.2f  with six fixed passes %5.2f  tail in %4.1f %% of the trips  density %.0f %%"
          % (name, passes.mean(), fixed6.mean(), 100 * (passes > 6).mean(), 100 * nw[:len(w) * 128].sum() / (fixed6.sum() * 256.0)))
# What they cost (csrc/scan.hip, the pass of round 5: 8 vector instructions per chain and pass, 24.8 issue cycles -- v_ffbl 4.2,
# v_lshrrev_b64 4.2, two shifts and two ands 9.6, v_lshlrev 2.4, v_cmp 4.4; tools/valu_rate.hip):
#  * a chain that may cross the middle of its word walks a 64-bit survivor mask: the 32-bit shift of the mask becomes a second
#    v_lshrrev_b64 (+1.8 cycles), and v_ffbl of the low dword alone must not run past an empty dword -- a v_min or a planted
#    sentinel per step (+2.4) --, and its 82 check bits (64 offsets + 18) no longer fit the 64-bit shift register: a third check
#    dword and a funnel shift per step (+4.2).  +8.4 cycles on 24.8 = +34 % per chain and pass for -13 % passes (6.99 -> 6.09).
#  * the top chain must stop at its quota (the middle survivor is not taken twice): a counter and a compare per step, or the
#    mask cut at the n/2-th survivor in the filter -- a select-k-th-bit, ~30 instructions per word, where the whole saving is
#    0.9 passes x 32 instructions per TWO words.
#  * the cousin keeps upward chains but its cut moves: a chain is then up to 44 offsets wide -- the same 64-bit mask.
# The kernel is bound by its vector instruction count (profiles/r06_scan: +5.4 % VALU = +10 % time, -17 % SALU bought nothing),
# so neither was built: both trade fewer passes for more instructions per pass at a loss.
