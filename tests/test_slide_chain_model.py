"""The arithmetic of the round-5 survivor pass of scan_slide_kernel (libbtbb_amd/csrc/scan.hip), modelled in numpy and held
against the straightforward form on the CPU -- what the GPU tests can only observe as "same hit list":

  * (the two-level form; the one-level form until the third session of round 6) a chain of 32 offsets as a pair of shift registers: the survivor mask and the 64-bit check register are moved down by
    v_ffbl of the mask, so the survivor in hand sits at bit 0 and its 19-bit index is the register's low 19 bits; a marker
    planted at bit 63 of the check register tells the offset by its distance from the top (v_ffbh of the high dword);
  * an exhausted chain shifts itself out (ffbl of 0 = -1: shift amounts 31 / 63) and indexes 0 or 1 ever after -- and 0
    and 1 are members of no candidate set (context.cpp refuses a table set where they are), so no "this lane has a
    survivor" term is needed;
  * the set as 16-bit entries, each bit-reversed, membership = the 16-bit sign of (entry << (index & 15)) -- sixteen bits since
    the third session of round 6: v_lshlrev_b16 issues at the fast rate on gfx950, v_lshlrev_b32 at the slow one.

Also the property trials_linear_kernel's arithmetic trial order rests on: the four bits that whiten a header's type field
take every value for exactly four of the 64 CLK1-6 candidates (packet.hip checks the same on the table it uploads)."""
import ctypes as C

import numpy as np

from _libs import oracle, seed

M64 = (1 << 64) - 1


def _ffbl(m):
    return (m & -m).bit_length() - 1 if m else 0xFFFFFFFF


def _brev32(x):
    return int(format(x, "032b")[::-1], 2)


def _walk(mask, c_lo, c_hi):
    """the kernel's passes over one chain: [(offset, index19)] in the order the kernel meets them, then two more passes"""
    m = mask
    reg = (((c_hi | 0x80000000) << 32) | c_lo) & M64
    out = []
    for _ in range(bin(mask).count("1") + 2):
        p = _ffbl(m)
        m >>= p & 31
        reg >>= p & 63
        hi = reg >> 32
        if m & 1:
            pos = 32 - hi.bit_length()                       # v_ffbh_u32 of the high dword
            out.append((pos, reg & 0x7FFFF))
        else:
            assert reg in (0, 1), "an exhausted chain must index 0 or 1"
            out.append((None, reg & 0x7FFFF))
        m &= ~1
    return out


def test_shift_register_chain_visits_every_survivor_with_its_index_and_offset():
    rng = np.random.default_rng(seed(5100))
    for it in range(4000):
        c_lo, c_hi = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
        dens = (0.0, 0.05, 0.125, 0.5, 1.0)[it % 5]
        mask = int(sum(1 << i for i in range(32) if rng.random() < dens))
        if it % 97 == 0:
            mask |= 1 << 31
        stream = (c_hi << 32) | c_lo                          # check bits 0 .. 63 of the chain
        got = _walk(mask, c_lo, c_hi)
        want = [(o, (stream >> o) & 0x7FFFF) for o in range(32) if (mask >> o) & 1]
        live = [g for g in got if g[0] is not None]
        assert live == want, (hex(mask), live[:4], want[:4])
        # (offset 31 needs check bits 31 .. 49: the marker at bit 63 is never among them)
        assert all(idx in (0, 1) for pos, idx in got if pos is None)


def _walk_abs(mask, c_lo, c_hi):
    """the one-level form since the third session of round 6: absolute positions, the check register untouched, no marker"""
    m = mask
    reg = ((c_hi << 32) | c_lo) & M64
    out = []
    for _ in range(bin(mask).count("1") + 2):
        p = _ffbl(m)                                         # 0xffffffff for an empty chain
        v = (reg >> (p & 63)) & 0xFFFFFFFF                   # v_lshrrev_b64 takes the low six bits of p
        if m:
            out.append((p & 31, v & 0x7FFFF))                # p is the offset a candidate event reports
        else:
            assert v in (0, 1), "an exhausted chain must index 0 or 1 (bit 63 of the register alone)"
            out.append((None, v & 0x7FFFF))
        m &= (m - 1) & 0xFFFFFFFF
    return out


def test_absolute_position_chain_visits_every_survivor_with_its_index_and_offset():
    rng = np.random.default_rng(seed(5102))
    for it in range(4000):
        c_lo, c_hi = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
        dens = (0.0, 0.05, 0.125, 0.5, 1.0)[it % 5]
        mask = int(sum(1 << i for i in range(32) if rng.random() < dens))
        if it % 97 == 0:
            mask |= 1 << 31
        stream = (c_hi << 32) | c_lo
        got = _walk_abs(mask, c_lo, c_hi)
        want = [(o, (stream >> o) & 0x7FFFF) for o in range(32) if (mask >> o) & 1]
        assert [g for g in got if g[0] is not None] == want, (hex(mask), got[:4], want[:4])
        assert all(idx in (0, 1) for pos, idx in got if pos is None)


def test_indices_0_and_1_are_in_no_candidate_set():
    import libbtbb_amd as bt
    lib = bt.lib()
    for n in (0, 1, 2, 3):
        words = (C.c_uint32 * (1 << 14))()
        members = lib.btbbx_slide_set(n, words, None)
        assert members > 0 and (words[0] & 3) == 0, (n, members, hex(words[0]))


def test_membership_as_the_sign_of_a_left_shift_of_the_bit_reversed_entry():
    import libbtbb_amd as bt
    lib = bt.lib()
    words = (C.c_uint32 * (1 << 14))()
    assert lib.btbbx_slide_set(2, words, None) == 1585
    plain = np.frombuffer(words, dtype=np.uint32)
    # the kernel's copy-in: every word bit-reversed, then its halves swapped = both 16-bit halves reversed in place
    lds = np.array([((_brev32(int(w)) >> 16) | (_brev32(int(w)) << 16)) & 0xFFFFFFFF for w in plain], dtype=np.uint32)
    entries = lds.view(np.uint16)                             # little-endian: entry k = half k & 1 of word k >> 1
    rng = np.random.default_rng(seed(5101))
    members = [i for i in range(1 << 19) if (int(plain[i >> 5]) >> (i & 31)) & 1]
    assert len(members) == 1585
    probe = members + [int(x) for x in rng.integers(0, 1 << 19, 20000)]
    for idx in probe:
        junk = int(rng.integers(0, 1 << 13)) << 19            # the register holds check bits above the index too
        v = idx | junk
        byte_off = (v >> 3) & 0xFFFE                          # the kernel's address (SET_BYTES - 2 = 0xfffe)
        assert byte_off % 2 == 0
        entry = int(entries[byte_off >> 1])
        sign = (((entry << (v & 15)) & 0xFFFF) >> 15) & 1     # v_lshlrev_b16 takes the low four bits of v; v_cmp_gt_i16 0, ...
        assert sign == ((int(plain[idx >> 5]) >> (idx & 31)) & 1), idx


def test_type_field_whitening_takes_every_value_for_four_clocks():
    orc = oracle()
    seen = {}
    for clk in range(64):
        zeros = (C.c_char * 18)()
        out = (C.c_char * 18)()
        # unwhiten of an all-zero header = the whitening bits themselves (bluetooth_packet.c:653-690); the reference's
        # only CLK1-6 choose the sequence position
        orc.orc_unwhiten(zeros, out, clk, 18, 0, 1)
        bits = [b & 1 for b in out.raw]
        seen.setdefault(sum(bits[3 + j] << j for j in range(4)), []).append(clk)
    assert sorted(seen) == list(range(16)) and all(len(v) == 4 for v in seen.values()), seen
