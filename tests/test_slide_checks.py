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
