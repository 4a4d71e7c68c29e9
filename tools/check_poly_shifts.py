#!/usr/bin/env python3
"""Which sliding check costs the LAP_ANY filter the fewest funnel shifts?  (round 6, CPU only)

scan_slide_kernel computes, per 64-offset word of a lane (dwords d0 d1 | d2 d3 = the word and its successor), the barker planes
(shifts 25 .. 31 of the pairs d2:d1 and d3:d2) and the check stream c = XOR of the stream shifted by the taps of a multiple q of the
reversed cofactor h~ (slide.h).  A tap t < 32 is a v_alignbit of pair (d[k+1], d[k]) by t, a tap t > 32 one of pair (d[k+2], d[k+1]) by
t - 32; equal (pair, shift) combinations are computed once (the compiler does that).  This enumerates every multiple of h~ of degree
<= 56 - SLIDE_BITS, at every anchor that keeps SLIDE_BITS checks inside window bits 1 .. 56, and counts the distinct funnel shifts per word.
"""
G = 0o260534236651
BITS = 19


def deg(p):
    return p.bit_length() - 1


def clmul(a, b):
    r = 0
    while b:
        low = b & -b
        r ^= a * low
        b ^= low
    return r


def cofactor(g):
    a, q = (1 << 63) | 1, 0
    while deg(a) >= deg(g):
        s = deg(a) - deg(g)
        q |= 1 << s
        a ^= g << s
    assert a == 0
    return q


def reversed_poly(p):
    d = deg(p)
    return sum(((p >> i) & 1) << (d - i) for i in range(d + 1))


def shifts_per_word(taps):
    T = [k for k in range(64) if (taps >> k) & 1]
    barker = set(range(25, 32))
    p0 = {t for t in T if 0 < t < 32}
    p1 = {t - 32 for t in T if t > 32} | {t for t in T if 0 < t < 32} | barker
    p2 = {t - 32 for t in T if t > 32} | barker
    return len(p0) + len(p1) + len(p2), T


hr = reversed_poly(cofactor(G))
print("h~: degree %d, weight %d" % (deg(hr), bin(hr).count("1")))
span = 56 - BITS
rows = []
for a in range(1, 2 << (span - deg(hr)), 2):
    q = clmul(hr, a)
    if deg(q) > span:
        continue
    for s in range(1, 56 - (BITS - 1) - deg(q) + 1):
        c, T = shifts_per_word(q << s)
        rows.append((c, len(T), s, T))
rows.sort()
print("funnel shifts per word (barker included) | taps | anchor | tap list -- the ten cheapest of %d candidates" % len(rows))
for r in rows[:10]:
    print(r)
print("in use (slide.h SLIDE_TAPS):", [r for r in rows if r[3] == [1, 3, 14, 19, 22, 23, 26, 28, 29, 34, 38]])
