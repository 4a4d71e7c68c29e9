"""The arithmetic of a wave-per-packet decoder for long packets (tests/_wave_model.py, NOTEBOOK.md 9 item 4) against the
oracle: per-lane unwhitening of 64-bit payload words, FEC 2/3 three blocks per lane with the 30-bit pieces gathered
into words, and the CRC as per-lane registers advanced by fixed matrices and XORed across the wave.  No kernel is
involved: this pins the algorithm (bit order, block alignment, the reference's early returns) before one is written."""
import numpy as np

import _libs
import _pkt
import _wave_model as wm
from libbtbb_amd import synth
from test_gpu_packets import _oracle_decode

DH = (synth.TYPE_DH1, synth.TYPE_AUX1, synth.TYPE_DH3, synth.TYPE_DH5)
DM = (synth.TYPE_DM1, synth.TYPE_DV, synth.TYPE_DM3, synth.TYPE_DM5)
MAXBODY = {synth.TYPE_DM1: 17, synth.TYPE_DH1: 27, synth.TYPE_DV: 9, synth.TYPE_AUX1: 29, synth.TYPE_DM3: 121,
           synth.TYPE_DH3: 183, synth.TYPE_DM5: 224, synth.TYPE_DH5: 339}


def test_wave_model_equals_the_oracle():
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(71))
    seen = {}
    for i in range(420):
        t = (DH + DM)[i % 8]
        lap, uap, clk6 = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 64))
        nb = MAXBODY[t] if i % 3 == 0 else int(rng.integers(0, MAXBODY[t] + 1))
        sym = synth.build_packet(lap, uap, clk6, t, lt_addr=int(rng.integers(0, 8)), flags=int(rng.integers(0, 8)),
                                 body=rng.integers(0, 256, nb, dtype=np.uint8).tobytes(), llid=int(rng.integers(0, 4)),
                                 flow=int(rng.integers(0, 2)), voice=rng.integers(0, 256, 10, dtype=np.uint8).tobytes())
        sym = np.concatenate([sym, rng.integers(0, 2, int(rng.integers(0, 200)), dtype=np.uint8)])[:3125]
        ne = int(rng.integers(0, 4)) if i % 2 else 0
        if ne:
            sym[rng.integers(126, len(sym), ne)] ^= 1            # payload region: the header stays decodable
        if i % 7 == 0:
            sym = sym[:int(rng.integers(130, len(sym) + 1))]      # capture cut short
        use_clk = clk6 if i % 5 else clk6 ^ 3                     # a wrong clock now and then: noise through every path
        sym = np.ascontiguousarray(sym)
        present, h, r, st = _oracle_decode(orc, sym, use_clk, uap)
        if not h:
            continue
        ptype = st["packet_type"]
        if ptype in (4, 9, 11, 15):
            rv, pay = wm.dh_wave(sym, use_clk & 63, uap, ptype)
        elif ptype in (3, 8, 10, 14):
            rv, pay = wm.dm_wave(sym, use_clk & 63, uap, ptype)
        else:
            continue
        assert rv == r, (i, ptype, rv, r, len(sym))
        if pay is not None and rv in (2, 10):
            assert np.array_equal(pay, st["payload"][:len(pay)]), (i, ptype, np.nonzero(pay != st["payload"][:len(pay)])[0][:8])
        seen[(ptype, rv)] = seen.get((ptype, rv), 0) + 1
    # every type decoded with a good CRC, FEC failures and short captures were all in the sample
    for ptype in (3, 4, 10, 11, 14, 15):
        assert seen.get((ptype, 10), 0) >= 5, (ptype, seen)
    assert sum(v for (t_, rv), v in seen.items() if rv == 0) >= 3 and sum(v for (t_, rv), v in seen.items() if rv == 1) >= 3, seen
    assert sum(v for (t_, rv), v in seen.items() if rv == 2) >= 10, seen


def test_lane_matrices_compose():
    """A^(64 l) applied to a block register = running the register over 64 l zero bits."""
    rng = np.random.default_rng(_libs.seed(72))
    adv = wm._adv64()
    for lane in (0, 1, 2, 17, 42, 63):
        reg = int(rng.integers(0, 1 << 16))
        assert wm._apply(adv[lane], reg) == wm._crc_step_bits(reg, np.zeros(64 * lane, np.uint8))


def test_three_words_per_lane_and_lane_prefix_registers():
    """The two CRC forms round 4 added to the kernel, on the CPU against the bit-serial register: three words per lane with
    A^(-192 l) (dh_payloads), and the register IN FRONT of every word from an XOR prefix over the lanes between two per-lane
    matrices (ev_payloads: that register, then eight byte steps, is how EV4 / EV5 find where their payload ends)."""
    rng = np.random.default_rng(_libs.seed(73))
    for trial in range(60):
        n_words = int(rng.integers(1, 44))
        uap = int(rng.integers(0, 256))
        words = [int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2)) for _ in range(n_words)]
        nbits = 64 * n_words
        bits = np.concatenate([wm._bits_of(w) for w in words])
        # registers in front of every word, serially (the seed is the register's start value)
        regs, reg = [], wm._seed(uap)
        for w in range(n_words):
            regs.append(reg)
            reg = wm._crc_step_bits(reg, bits[64 * w:64 * w + 64])
        # (lane 0 starts from zero: the seed rides on the first sixteen bits of word 0, where the kernel's byte steps meet it)
        assert wm.ev_registers_by_lane_prefix(words, uap) == [0] + regs[1:], trial
        # make the payload's CRC come out right now and then: the last sixteen bits = the register in front of them
        if trial % 2:
            tail = wm._crc_step_bits(wm._seed(uap), bits[:nbits - 16])
            words[-1] = (words[-1] & ((1 << 48) - 1)) | (tail << 48)
            bits = np.concatenate([wm._bits_of(w) for w in words])
        want = wm._crc_step_bits(wm._seed(uap), bits) == 0
        assert want == bool(trial % 2) or not trial % 2, trial
        assert wm.wave_crc_is_zero_three_words_per_lane(words, nbits, uap) == want, trial
        assert wm.wave_crc_is_zero_two_words_per_lane(words, nbits, uap) == want, trial
        assert wm.wave_crc_is_zero_start_aligned(words, nbits, uap) == want, trial


def test_four_fec23_blocks_per_lane():
    """long_payloads' FEC step the way the kernel indexes it -- three dwords, two funnel shifts, the third block across the
    seam, blocks behind the packet's last masked to zero -- against one block at a time, at every bit alignment, with symbol
    errors; and where the 40 bits go: byte 5 n of the packed payload."""
    rng = np.random.default_rng(_libs.seed(74))
    for trial in range(200):
        n_blocks = int(rng.integers(1, 30))
        start = int(rng.integers(0, 97))
        data = rng.integers(0, 2, 10 * n_blocks, dtype=np.uint8)
        coded = synth.fec23_encode(data) if hasattr(synth, "fec23_encode") else None
        if coded is None:
            coded = np.concatenate([np.concatenate([data[10 * b:10 * b + 10], _parity(data[10 * b:10 * b + 10])]) for b in range(n_blocks)])
        stream = np.concatenate([rng.integers(0, 2, start, dtype=np.uint8), coded, rng.integers(0, 2, 200, dtype=np.uint8)])
        for _ in range(int(rng.integers(0, 4))):
            stream[start + int(rng.integers(0, 15 * n_blocks))] ^= 1
        packed, any_bad = 0, False
        for n in range((n_blocks + 3) // 4):
            have = min(4, n_blocks - 4 * n)
            got, bad = wm.fec23_quad(stream, start + 60 * n, have)
            any_bad |= bad
            packed |= got << (8 * 5 * n)                      # byte 5 n
        want, want_bad = 0, False
        for b in range(n_blocks):
            ok, d = wm._fec23_block(stream[start + 15 * b:start + 15 * b + 15])
            want_bad |= not ok
            want |= wm._int_of(d) << (10 * b)
        assert any_bad == want_bad, trial
        assert packed == want, trial


def _parity(d10):
    par = 0
    for i in range(10):
        if d10[i]:
            par ^= wm._F23[i]
    return np.array([(par >> k) & 1 for k in range(5)], dtype=np.uint8)
