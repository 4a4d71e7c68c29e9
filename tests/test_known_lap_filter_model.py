"""The arithmetic of scan_known_lap_kernel's bit-sliced filter (libbtbb_amd/csrc/scan.hip, second half of round 6), modelled in numpy
and held against its definition on the CPU -- what the GPU tests can only observe as "same hit list":

  * the filter counts mismatches in sync-word bits 24..31 and 56..63 (sixteen planes; twelve -- 28..31 and 56..63 -- for limits 0 / 1);
  * plane j of a dword pair (hi : lo) = the funnel shift of the pair by 24 + j (v_alignbit); for the 32 offsets p of a half that starts
    at dword k it is window bit 24 + j taken from the pair (k + 1 : k) and window bit 56 + j taken from the pair (k + 2 : k + 1) -- so the
    UPPER planes of half k are the LOWER planes of half k + 1 (the kernel computes every set once: pair_planes), along a lane's run of
    two consecutive words + the word behind them (six dwords, four halves, five plane sets);
  * count <= limit over those planes is a necessary condition of find_known_lap's popcount(window ^ syncword) <= limit
    (bluetooth_packet.c:433), never a sufficient one: every oracle hit passes the model, the model passes more."""
import numpy as np

from _libs import LAP_ANY, oracle, orc_find_all, seed
from libbtbb_amd import synth

M32 = 0xFFFFFFFF


def _alignbit(hi, lo, sh):
    return ((((hi & M32) << 32) | (lo & M32)) >> sh) & M32


def _pair_planes(lo, hi):
    return [_alignbit(hi, lo, 24 + j) for j in range(8)]


def _count_le(lowp, highp, sync, limit, first_low):
    """32-bit mask of the offsets of a half whose mismatch count in the chosen sync-word bits is <= limit"""
    out = 0
    for p in range(32):
        n = 0
        for j in range(first_low, 8):
            n += ((lowp[j] >> p) & 1) ^ ((sync >> (24 + j)) & 1)
        for j in range(8):
            n += ((highp[j] >> p) & 1) ^ ((sync >> (56 + j)) & 1)
        out |= (1 if n <= limit else 0) << p
    return out


def _model_survivors(dwords, sync, limit):
    """the kernel's filter over a stream given as dwords: lanes own runs of two words (four halves), the planes of a pair are
    computed once and used by both halves they belong to -> set of surviving offsets"""
    first_low = 0 if limit >= 2 else 4
    surv = set()
    n_halves = len(dwords) - 2
    for run in range(0, n_halves, 4):                       # a lane's run: halves run .. run + 3, dwords run .. run + 5
        D = [dwords[run + k] if run + k < len(dwords) else 0 for k in range(6)]
        sets = [_pair_planes(D[k], D[k + 1]) for k in range(5)]          # five plane sets for four halves
        for c in range(4):
            if run + c >= n_halves:
                break
            m = _count_le(sets[c], sets[c + 1], sync, limit, first_low)
            surv |= {32 * (run + c) + p for p in range(32) if (m >> p) & 1}
    return surv


def test_the_shared_planes_are_the_window_bits_they_stand_for():
    rng = np.random.default_rng(seed(5))
    d = [int(x) for x in rng.integers(0, 1 << 32, 7, dtype=np.uint64)]
    stream = 0
    for k, v in enumerate(d):
        stream |= v << (32 * k)
    for k in range(4):                                       # half k: offsets 32 k .. 32 k + 31
        low, high = _pair_planes(d[k], d[k + 1]), _pair_planes(d[k + 1], d[k + 2])
        for p in (0, 1, 7, 8, 15, 24, 30, 31):
            window = (stream >> (32 * k + p)) & ((1 << 64) - 1)
            for j in range(8):
                assert (low[j] >> p) & 1 == (window >> (24 + j)) & 1
                assert (high[j] >> p) & 1 == (window >> (56 + j)) & 1
    # (so the set of the pair (k + 2 : k + 1) serves half k as its upper planes and half k + 1 as its lower ones: checked above for both)


def test_filter_model_is_a_necessary_condition_and_equals_its_definition():
    orc = oracle()
    orc.orc_reset_syndrome_map()
    orc.orc_init(2)
    for lap in (0x9E8B33, 0x1E8B33):
        sync = synth.syncword(lap)
        words, _ = synth.make_stream(seed(77), 512, stride=512, lap=lap)
        sym = np.ascontiguousarray(synth.unpack_bits(words))
        dwords = [int(x) for x in words.view(np.uint32)]
        n = len(sym) - 63
        for limit in (0, 1, 2, 3):
            model = _model_survivors(dwords, sync, limit)
            # the definition: mismatches of the window in the chosen bits
            chosen = [b for b in range(64) if (24 + (0 if limit >= 2 else 4) <= b <= 31) or b >= 56]
            sbits = np.array([(sync >> b) & 1 for b in range(64)], dtype=np.uint8)
            cnt = np.zeros(n, dtype=np.int32)
            for b in chosen:
                cnt += sym[b:b + n] != sbits[b]
            want = set(np.nonzero(cnt <= limit)[0].tolist())
            assert {o for o in model if o < n} == want
            hits = {o for (o, _, _) in orc_find_all(sym, n, lap, limit)}
            assert hits <= want and len(hits) >= 10
            if limit >= 2:
                assert len(want) < n // 50                                # sixteen planes: a sparse survivor set


def test_limit_2_and_3_network_of_the_sixteen_plane_filter_equals_the_count():
    """top16_filter's shortcut for limits 2 and 3 (third session of round 6), over all 2^16 values of the sixteen mismatch bits:
    five full adders over m0 .. m14, two over their sums and m15, then
        count <= 2  <=>  at_most_one(W) & ~(any(W) & (o1 | o2)),   count <= 3  <=>  at_most_one(W) & ~(any(W) & o1 & o2)
    with W = c0 .. c4, k0, k1 in the groups (c0 c1 c2) (c3 c4 k0) (k1)."""
    v = np.arange(1 << 16, dtype=np.uint32)
    m = [((v >> k) & 1).astype(bool) for k in range(16)]
    fa = lambda a, b, c: (a ^ b ^ c, (a & b) | (a & c) | (b & c))             # noqa: E731
    s0, c0 = fa(m[0], m[1], m[2])
    s1, c1 = fa(m[3], m[4], m[5])
    s2, c2 = fa(m[6], m[7], m[8])
    s3, c3 = fa(m[9], m[10], m[11])
    s4, c4 = fa(m[12], m[13], m[14])
    o1, k0 = fa(s0, s1, s2)
    o2, k1 = fa(s3, s4, m[15])
    a0, t0 = c0 | c1 | c2, fa(c0, c1, c2)[1]
    a1, t1 = c3 | c4 | k0, fa(c3, c4, k0)[1]
    two_groups, any_w = fa(a0, a1, k1)[1], a0 | a1 | k1
    bad = t0 | t1 | two_groups
    count = sum(x.astype(np.int32) for x in m)
    assert np.array_equal(~(bad | (any_w & (o1 | o2))), count <= 2)
    assert np.array_equal(~(bad | (any_w & o1 & o2)), count <= 3)


def test_limit_0_and_1_networks_of_the_twelve_plane_filter_equal_the_count():
    """top12_filter's shortcuts (third session of round 6), over all 2^12 values of the twelve mismatch bits: limit 0 = none of them;
    limit 1: four full adders over m0 .. m11, one over their sums s0 s1 s2 -> (o1, k0); count = o1 + s3 + 2 x (c0 .. c3, k0)."""
    v = np.arange(1 << 12, dtype=np.uint32)
    m = [((v >> k) & 1).astype(bool) for k in range(12)]
    fa = lambda a, b, c: (a ^ b ^ c, (a & b) | (a & c) | (b & c))             # noqa: E731
    (s0, c0), (s1, c1), (s2, c2), (s3, c3) = fa(m[0], m[1], m[2]), fa(m[3], m[4], m[5]), fa(m[6], m[7], m[8]), fa(m[9], m[10], m[11])
    o1, k0 = fa(s0, s1, s2)
    count = sum(x.astype(np.int32) for x in m)
    assert np.array_equal(~((c0 | c1 | c2) | c3 | k0 | (o1 & s3)), count <= 1)
    any_m = m[0]
    for x in m[1:]:
        any_m = any_m | x
    assert np.array_equal(~any_m, count == 0)
