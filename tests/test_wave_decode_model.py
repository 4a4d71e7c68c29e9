"""The arithmetic of a wave-per-packet decoder for long packets (tests/_wave_model.py) against the
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
