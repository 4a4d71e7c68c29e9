"""Round 5's last two changes to scan_slide_kernel (libbtbb_amd/csrc/scan.hip), restated in numpy and held against their definitions on
the CPU -- the GPU tests see them only as "same hit list":

  * the tile geometry: a wave owns 63 words of a tile (12 x 63 = 756 words, 16 x 63 = 1008 in the two-level form), its lane 63 works on the
    next wave's first word and has no offsets of its own; the launcher calls a tile "full" when every lane's two loads and every live
    lane's 64 offsets are in range;
  * the two-level form for tables of three and four errors: the chains run on the COMPLEMENTED first check stream (an idle chain then
    indexes 0 or 1, which btbbx_slide_sets_two_level keeps out of the set), a member's look-up in the second set takes 24 positions of the
    second check stream at the offset the chain's marker gives.  Modelled: shift-register chains over random streams with planted sync
    words find exactly the offsets whose checks, taken by definition from the window, are members of both sets."""
import ctypes as C

import numpy as np
import pytest

from _libs import oracle, seed

M64 = (1 << 64) - 1


@pytest.mark.parametrize("waves", [12, 16])
def test_waves_of_63_words_cover_every_offset_once_and_full_tiles_stay_in_range(waves):
    tile_words = waves * 63
    rng = np.random.default_rng(seed(5200 + waves))
    for case in range(300):
        n_words = int(rng.integers(2, 6 * tile_words))
        search_bits = int(rng.integers(1, n_words * 64 - 63 + 1))
        search_words = (search_bits + 63) // 64
        tiles = (search_words + tile_words - 1) // tile_words
        # launch_scan: tile t is full iff (t + 1) * tile_words + 2 <= n_words and (t + 1) * tile_words * 64 <= search_bits
        by_words = (n_words - 2) // tile_words if n_words >= 2 else 0
        full_tiles = min(by_words, search_bits // (tile_words * 64))
        owner = np.zeros(search_bits, dtype=np.uint8)
        for t in range(tiles):
            for w in range(waves):
                for lane in (0, 1, 31, 62, 63):               # (the lanes between behave like lane 1 .. 62)
                    word = t * tile_words + w * 63 + lane
                    if t < full_tiles:
                        assert word + 1 < n_words, (case, t, w, lane)          # both loads of every lane, lane 63 included
                        if lane != 63:
                            assert word * 64 + 63 < search_bits
                    # code_word(): what a candidate's position code says (tile iteration, wave, lane, offset in the word)
                    assert t * tile_words + w * 63 + lane == word
        # every word below search_words belongs to exactly one (tile, wave, lane != 63)
        words = np.arange(search_words)
        t, r = words // tile_words, words % tile_words
        w, lane = r // 63, r % 63
        assert (t < tiles).all() and (w < waves).all() and (lane < 63).all()
        back = t * tile_words + w * 63 + lane
        assert (back == words).all()
        # ... and lane 63 of a wave is lane 0 of the next wave (or of the next tile's first wave): the same word, owned there
        assert ((t * tile_words + w * 63 + 63) == (t * tile_words + (w + 1) * 63)).all()


def _bits(v, n):
    return [(v >> i) & 1 for i in range(n)]


def _check_stream(stream_bits, taps, n):
    """c(x) = XOR of stream[x + k] over the taps k, x = 0 .. n - 1"""
    ks = [k for k in range(64) if (taps >> k) & 1]
    out = np.zeros(n, dtype=np.uint8)
    for k in ks:
        out ^= stream_bits[k:k + n]
    return out


@pytest.mark.parametrize("n_err", [3, 4])
def test_two_level_chain_walk_finds_what_the_definition_finds(n_err):
    import libbtbb_amd as bt
    lib = bt.lib()
    orc = oracle()
    first = (C.c_uint32 * (1 << 15))()
    second = (C.c_uint32 * (1 << 19))()
    taps = (C.c_uint64 * 2)()
    assert lib.btbbx_slide_sets_two_level(n_err, first, second, taps) == 0
    f = np.frombuffer(first, dtype=np.uint32)
    s = np.frombuffer(second, dtype=np.uint32)
    ta, tb = int(taps[0]), int(taps[1])
    rng = np.random.default_rng(seed(5210 + n_err))
    n_words = 40
    nbits = n_words * 64
    stream = rng.integers(0, 2, nbits + 128, dtype=np.uint8)
    planted = []
    for k in range(12):                                            # sync words with 0 .. n_err errors below the barker bits
        off = 200 * k + int(rng.integers(0, 100))                    # (apart: no planted window overwrites another)
        w = orc.orc_gen_syncword(int(rng.integers(0, 1 << 24)))
        for e in rng.choice(57, size=k % (n_err + 1), replace=False):
            w ^= 1 << int(e)
        stream[off:off + 64] = _bits(w, 64)
        planted.append(off)
    ca = _check_stream(stream, ta, nbits + 32) ^ 1                 # the kernel's first stream, complemented
    cb = _check_stream(stream, tb, nbits + 32)
    surv = rng.random(nbits) < 0.125                               # any survivor mask will do: the filter is not the subject here
    surv[planted] = True

    def member1(idx):
        return (int(f[idx >> 5]) >> (idx & 31)) & 1

    def member2(idx):
        return (int(s[idx >> 5]) >> (31 - (idx & 31))) & 1

    # by definition, offset by offset
    want = []
    for o in np.flatnonzero(surv):
        i1 = sum(int(ca[o + j]) << j for j in range(20))
        i2 = sum(int(cb[o + j]) << j for j in range(24))
        if member1(i1) and member2(i2):
            want.append(int(o))
    assert set(planted) <= set(want)                               # every planted window is within n_err errors of a sync word

    # the kernel's way: chains of 32 offsets as shift registers, marker at bit 63, idle chains read on
    got = []
    for base in range(0, nbits, 32):
        m = sum(int(surv[base + j]) << j for j in range(32))
        reg = sum(int(ca[base + j]) << j for j in range(64))
        reg = (reg & ~(1 << 63) | (1 << 63)) & M64                  # (c[h + 1] | 0x80000000) : c[h]
        b_lo = sum(int(cb[base + j]) << j for j in range(64))       # c2[h + 1] : c2[h]
        for _ in range(bin(m).count("1") + 2):
            p = ((m & -m).bit_length() - 1) if m else 0xFFFFFFFF
            m >>= p & 31
            reg >>= p & 63
            idx = reg & 0xFFFFF
            if member1(idx):
                assert m & 1, "an idle chain must never be a member"
                pos = 32 - (reg >> 32).bit_length()                 # v_ffbh of the high dword
                v2 = (b_lo >> pos) & 0xFFFFFFFF                     # v_alignbit(c2[h + 1], c2[h], pos)
                if member2(v2 & 0xFFFFFF):
                    got.append(base + pos)
            m &= ~1
    assert sorted(got) == want
