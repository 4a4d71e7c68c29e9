#!/usr/bin/env python3
"""How much could a SECOND bit-sliced filter, in front of the survivor passes of scan_slide_kernel, remove?

The survivor passes look 1/8 of the offsets (the barker survivors) up one by one in the 2^19-bit set of csrc/slide.h; the launch is
bound by those vector instructions.  A bit-sliced test over m planes of the check stream (one funnel shift each, 4.15
cycles) that is true for every member of the set and false for most other values would thin the survivors out before the passes.
This script bounds what ANY such test can do, from the set itself:

  * a test over the planes J can at best pass |projection of the set onto J| / 2^|J| of the random survivors;
  * the cheap family -- "at most 2r of the planes J are violated", r = the most checks of J one bit error can flip -- needs checks
    with (nearly) disjoint supports, i.e. shifts j, j' of the tap pattern whose difference is not a difference of two taps.

Both come out empty for the check in use (and it is the lightest of its degree, tools/check_poly_shifts.py): every difference 1..18
but one is a difference of two taps, and a projection onto fewer than ten planes passes more than three quarters of all values.
No GPU needed.  Reference: the set is the one bluetooth_packet.c:161-185 (gen_syndrome_map) implies, restricted to the 19 checks
inside window bits 1..56.
"""
import itertools
import random

GEN = 0o260534236651          # generator of the (64,30) code's cyclic part (common.h SW_POLY), as in csrc/slide.h
SLIDE_BITS, SPAN = 19, 56 - 19


def degree(p):
    return p.bit_length() - 1


def clmul(a, b):
    r = 0
    while b:
        r ^= a << ((b & -b).bit_length() - 1)
        b &= b - 1
    return r


def lightest_check(g, span):
    a, q = (1 << 63) | 1, 0
    while degree(a) >= degree(g):
        s = degree(a) - degree(g)
        q |= 1 << s
        a ^= g << s
    assert a == 0
    hr = int(bin(q)[2:][::-1], 2)
    best = hr
    for m in range(1, 2 << (span - degree(hr)), 2):
        c = clmul(hr, m)
        if degree(c) <= span and bin(c).count("1") < bin(best).count("1"):
            best = c
    return best


def main():
    taps_poly = lightest_check(GEN, SPAN) << 1
    taps = [k for k in range(64) if (taps_poly >> k) & 1]
    print("taps of the check stream (SLIDE_TAPS):", taps)
    tset = set(taps)
    diffs = {b - a for a in taps for b in taps if b > a}
    print("differences 1..18 that are NOT a difference of two taps:", [d for d in range(1, SLIDE_BITS) if d not in diffs])
    fam = max((J for r in range(1, 5) for J in itertools.combinations(range(SLIDE_BITS), r)
               if all((b - a) not in diffs for a, b in itertools.combinations(J, 2))), key=len)
    print("largest family of checks with pairwise disjoint supports:", fam,
          "-> 'at most two of them violated' passes %d / %d of all values" % (1 + len(fam) + len(fam) * (len(fam) - 1) // 2, 2 ** len(fam)))
    # error patterns: <= 2 flipped bits among window bits 0..57 (bluetooth_packet.c:167); check j covers bits j + tap
    cols = [sum(1 << j for j in range(SLIDE_BITS) if (p - j) in tset) for p in range(58)]
    members = {0} | set(cols) | {a ^ b for a, b in itertools.combinations(cols, 2)}
    print("members of the 2^19-bit set for two errors:", len(members), "(%.2f %% of all values)" % (100.0 * len(members) / 2 ** SLIDE_BITS))
    rng = random.Random(2)
    print("threshold tests: m planes, r = most planes one bit error flips, share of random values with <= 2r violated planes")
    for m in (6, 8, 10, 12):
        best_r, best_j = 99, None
        for _ in range(20000):
            J = rng.sample(range(SLIDE_BITS), m)
            r = max(sum(1 for j in J if (p - j) in tset) for p in range(58))
            if r < best_r:
                best_r, best_j = r, sorted(J)
        share = sum(1 for v in range(2 ** m) if bin(v).count("1") <= 2 * best_r) / 2 ** m
        print("  m = %2d: r = %d (planes %s): passes %.1f %%" % (m, best_r, best_j, 100 * share))
    print("any test at all over m planes: smallest |projection of the set| / 2^m found")
    for m in (4, 6, 8, 10, 12):
        cands = itertools.combinations(range(SLIDE_BITS), m) if m <= 6 else \
            (tuple(sorted(rng.sample(range(SLIDE_BITS), m))) for _ in range(4000))
        best, best_j = 2.0, None
        for J in cands:
            n = len({sum(((x >> j) & 1) << i for i, j in enumerate(J)) for x in members})
            if n / 2 ** m < best:
                best, best_j = n / 2 ** m, J
        print("  m = %2d: >= %.1f %% pass (planes %s; %d distinct projections = minterms of the test)" % (m, 100 * best, best_j, round(best * 2 ** m)))


if __name__ == "__main__":
    main()
