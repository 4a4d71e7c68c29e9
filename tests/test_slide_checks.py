"""The sliding parity checks of the LAP_ANY survivor loop (libbtbb_amd/csrc/slide.h) against the oracle's
sync words: every check -- the compile-time tap pattern shifted to 0 .. SLIDE_BITS-1 -- must have even parity
on (sync word ^ PN) for every LAP, and must not touch the seven bits the barker correction replaces
(bluetooth_packet.c:387-399).  CPU only: the header is compiled into a tiny host program."""
import os
import subprocess

import numpy as np

from _libs import ROOT, oracle, seed

PN = 0x83848D96BBCC54FC      # bluetooth_packet.c:115


def _slide_constants(tmp_path):
    src = tmp_path / "slide_print.cpp"
    src.write_text('#include <stdio.h>\n#include "%s"\n'
                   'int main() { printf("%%llx %%d\\n", (unsigned long long)SLIDE_TAPS, SLIDE_BITS); return 0; }\n'
                   % os.path.join(ROOT, "libbtbb_amd", "csrc", "slide.h"))
    exe = tmp_path / "slide_print"
    subprocess.run(["g++", "-std=c++17", "-O1", str(src), "-o", str(exe)], check=True)
    taps, bits = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return int(taps, 16), int(bits)


def test_checks_vanish_on_every_sync_word(tmp_path):
    taps, bits = _slide_constants(tmp_path)
    assert taps & 1 == 0, "check 0 of a length-64 codeword does not hold: the pattern starts at bit 1"
    assert (taps << (bits - 1)) >> 57 == 0, "the checks must stay below the barker bits"
    orc = oracle()
    rng = np.random.default_rng(seed(4100))
    laps = [0, 0xFFFFFF, 0x9E8B33, 0x123456] + [int(x) for x in rng.integers(0, 1 << 24, 4000)]
    for lap in laps:
        cw = orc.orc_gen_syncword(lap) ^ PN
        for b in range(bits):
            assert bin(cw & (taps << b)).count("1") % 2 == 0, (hex(lap), b)


def test_index_of_a_damaged_sync_word_is_a_sum_of_columns(tmp_path):
    """What the bitmap in context.cpp relies on: errors in bits 0..56 move the index by the XOR of their columns,
    whatever stands in bits 57..63."""
    taps, bits = _slide_constants(tmp_path)
    orc = oracle()
    rng = np.random.default_rng(seed(4101))

    def index(window):
        return sum((bin(window & (taps << b)).count("1") & 1) << b for b in range(bits))

    k_pn = index(PN)
    col = [sum((((taps >> (i - b)) & 1) if i >= b else 0) << b for b in range(bits)) for i in range(57)]
    for _ in range(3000):
        lap = int(rng.integers(0, 1 << 24))
        errs = rng.choice(57, size=int(rng.integers(0, 4)), replace=False)
        window = orc.orc_gen_syncword(lap)
        expect = k_pn
        for e in errs:
            window ^= 1 << int(e)
            expect ^= col[int(e)]
        window ^= int(rng.integers(0, 128)) << 57
        assert index(window) == expect


def test_product_candidate_set_contains_every_acceptable_window(tmp_path):
    """btbbx_slide_set (host only) is the set the kernel loads into LDS: every window the reference's rule can accept --
    a sync word with at most n errors in bits 0..56 and anything in the seven bits the barker correction replaces --
    must be a member, and the set must be no larger than the patterns allow."""
    import ctypes as C
    from math import comb

    import libbtbb_amd as bt

    lib = bt.lib()                      # loads without a GPU; this entry point makes no HIP call
    taps_c, bits = _slide_constants(tmp_path)
    orc = oracle()
    rng = np.random.default_rng(seed(4102))
    for n in (0, 1, 2, 3):
        words = (C.c_uint32 * (1 << (bits - 5)))()
        taps = C.c_uint64(0)
        members = lib.btbbx_slide_set(n, words, C.byref(taps))
        assert taps.value == taps_c
        assert 1 <= members <= sum(comb(57, k) for k in range(n + 1))
        bitmap = np.frombuffer(words, dtype=np.uint32)
        assert int(np.unpackbits(bitmap.view(np.uint8)).sum()) == members
        for _ in range(1500):
            window = orc.orc_gen_syncword(int(rng.integers(0, 1 << 24)))
            for e in rng.choice(57, size=int(rng.integers(0, n + 1)), replace=False):
                window ^= 1 << int(e)
            window ^= int(rng.integers(0, 128)) << 57
            idx = sum((bin(window & (taps_c << b)).count("1") & 1) << b for b in range(bits))
            assert (int(bitmap[idx >> 5]) >> (idx & 31)) & 1, (n, hex(window))
    # two errors: the distinct XORs of at most two columns, rebuilt here from the tap pattern
    col = [sum((((taps_c >> (i - b)) & 1) if i >= b else 0) << b for b in range(bits)) for i in range(57)]
    distinct = {0} | set(col) | {col[i] ^ col[j] for i in range(57) for j in range(i)}
    assert lib.btbbx_slide_set(2, words, None) == len(distinct)


def _two_level_constants(tmp_path):
    src = tmp_path / "slide4_print.cpp"
    src.write_text('#include <stdio.h>\n#include "%s"\n'
                   'int main() { printf("%%llx %%d %%llx %%d\\n", (unsigned long long)SLIDE4_TAPS, SLIDE4_BITS, '
                   '(unsigned long long)SLIDE4B_TAPS, SLIDE4B_BITS); return 0; }\n'
                   % os.path.join(ROOT, "libbtbb_amd", "csrc", "slide.h"))
    exe = tmp_path / "slide4_print"
    subprocess.run(["g++", "-std=c++17", "-O1", str(src), "-o", str(exe)], check=True)
    t1, b1, t2, b2 = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return int(t1, 16), int(b1), int(t2, 16), int(b2)


def test_two_level_checks_vanish_on_every_sync_word_and_span_the_checks_below_the_barker_bits(tmp_path):
    """The two check streams of the kernel for three and four errors (slide.h: SLIDE4_TAPS, SLIDE4B_TAPS): every position of both
    has even parity on every (sync word ^ PN) and stays inside bits 1 .. 56; together they have rank 27 -- all there is: the
    26 .. 27 independent checks of the code that avoid bit 0 and the seven bits the barker correction replaces."""
    t1, b1, t2, b2 = _two_level_constants(tmp_path)
    assert (b1, b2) == (20, 24) and t1 != t2
    orc = oracle()
    rng = np.random.default_rng(seed(4110))
    for taps, bits in ((t1, b1), (t2, b2)):
        assert taps & 1 == 0 and (taps << (bits - 1)) >> 57 == 0
        for lap in [0, 0xFFFFFF, 0x9E8B33] + [int(x) for x in rng.integers(0, 1 << 24, 1500)]:
            cw = orc.orc_gen_syncword(lap) ^ PN
            assert all(bin(cw & (taps << b)).count("1") % 2 == 0 for b in range(bits)), hex(lap)
    rows = [t1 << b for b in range(b1)] + [t2 << b for b in range(b2)]
    basis = []
    for v in rows:
        for x in basis:
            v = min(v, v ^ x)
        if v:
            basis.append(v)
    assert len(basis) == 27


def test_two_level_sets_contain_every_acceptable_window_and_no_idle_chain(tmp_path):
    """btbbx_slide_sets_two_level hands out the sets as the kernel reads them (first: indexed by the complemented checks, plain bit
    order; second: bit-reversed words).  Every window within n errors of a sync word (anything in the barker bits) is a member of
    both; index 0 and 1 of the first -- what an idle chain of the kernel reads -- are not; and the sets are as selective as
    DESIGN 3.1 says (a third / 3 % of all values for four / three errors in the first, a few per cent in the second)."""
    import ctypes as C

    import libbtbb_amd as bt

    lib = bt.lib()
    t1, b1, t2, b2 = _two_level_constants(tmp_path)
    orc = oracle()
    rng = np.random.default_rng(seed(4111))
    first = (C.c_uint32 * (1 << (b1 - 5)))()
    second = (C.c_uint32 * (1 << (b2 - 5)))()
    taps = (C.c_uint64 * 2)()
    assert lib.btbbx_slide_sets_two_level(2, first, second, taps) < 0
    for n, lo1, hi1, hi2 in ((3, 0.02, 0.04, 0.003), (4, 0.28, 0.34, 0.03)):
        assert lib.btbbx_slide_sets_two_level(n, first, second, taps) == 0
        assert (taps[0], taps[1]) == (t1, t2)
        f = np.frombuffer(first, dtype=np.uint32)
        s = np.frombuffer(second, dtype=np.uint32)
        assert (int(f[0]) & 3) == 0
        d1 = np.unpackbits(f.view(np.uint8)).mean()
        d2 = np.unpackbits(s.view(np.uint8)).mean()
        assert lo1 < d1 < hi1 and 0 < d2 < hi2, (n, d1, d2)
        for _ in range(1500):
            window = orc.orc_gen_syncword(int(rng.integers(0, 1 << 24)))
            for e in rng.choice(57, size=int(rng.integers(0, n + 1)), replace=False):
                window ^= 1 << int(e)
            window ^= int(rng.integers(0, 128)) << 57
            i1 = sum((bin(window & (t1 << b)).count("1") & 1) << b for b in range(b1)) ^ ((1 << b1) - 1)
            i2 = sum((bin(window & (t2 << b)).count("1") & 1) << b for b in range(b2))
            assert (int(f[i1 >> 5]) >> (i1 & 31)) & 1, (n, hex(window))
            assert (int(s[i2 >> 5]) >> (31 - (i2 & 31))) & 1, (n, hex(window))


def test_front_set_of_the_five_error_tables_contains_every_acceptable_window(tmp_path):
    """btbbx_slide_sets_two_level(5, ...) hands out the 2^24-bit set scan_lap_any_kernel probes first when the tables are built for
    five errors (round 6; bit-reversed words, the layout of the two-level form's second set): every window within five errors of a
    sync word (anything in the barker bits) is a member, and 22.4 % of all values are (3 756 016 members: what lets four survivors in
    five skip the 8 MiB bitmap over the syndrome)."""
    import ctypes as C

    import libbtbb_amd as bt

    lib = bt.lib()
    _, _, t2, b2 = _two_level_constants(tmp_path)
    orc = oracle()
    rng = np.random.default_rng(seed(4115))
    second = (C.c_uint32 * (1 << (b2 - 5)))()
    taps = (C.c_uint64 * 2)()
    assert lib.btbbx_slide_sets_two_level(5, None, second, taps) == 0
    assert taps[1] == t2
    s = np.frombuffer(second, dtype=np.uint32)
    assert int(np.unpackbits(s.view(np.uint8)).sum()) == 3756016
    for _ in range(3000):
        window = orc.orc_gen_syncword(int(rng.integers(0, 1 << 24)))
        for e in rng.choice(57, size=int(rng.integers(0, 6)), replace=False):
            window ^= 1 << int(e)
        window ^= int(rng.integers(0, 128)) << 57
        i2 = sum((bin(window & (t2 << b)).count("1") & 1) << b for b in range(b2))
        assert (int(s[i2 >> 5]) >> (31 - (i2 & 31))) & 1, hex(window)
