"""GPU parity for hop selection and the CLK1-27 reversal (hop.hip, piconet.cpp) against the
oracle's materialised 2^27-entry patterns, against the fixture recorded from the reference
(tests/golden/hop.json), and end to end through btbb_process_packet."""
import ctypes as C
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import _hop
import _libs
import libbtbb_amd as bt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
HOP = json.load(open(os.path.join(HERE, "golden", "hop.json")))


@pytest.fixture(scope="module", autouse=True)
def ready():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    bt.init(2)
    _libs.oracle().orc_init(2)
    yield
    _libs.oracle().orc_hop_cache_clear()


def _amap(case):
    return None if case["afh_map"] is None else np.array(case["afh_map"], np.uint8)


@pytest.mark.parametrize("case", HOP["cases"], ids=lambda c: "%06x" % c["lap"])
def test_whole_pattern_matches_reference_fixture(case):
    seq = bt.hop_sequence(bt.hop_cfg(case["lap"], case["uap"], _amap(case)))
    assert seq[:512].tolist() == case["head"]
    assert [zlib.crc32(seq[i << 20:(i + 1) << 20].tobytes()) for i in range(128)] == case["crc32_per_mib"]
    assert hashlib.sha256(seq.tobytes()).hexdigest() == case["sha256"]


def test_pattern_slices_and_single_clocks_match_oracle():
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(61))
    for lap, uap, used in ((0x31337A, 0x9C, None), (0x5A5A5A, 0x01, 23), (0x000000, 0x00, 79), (0xFFFFFF, 0xFF, 1)):
        amap = _hop.afh_map_bytes(rng, used) if used else None
        pn, want = _hop.orc_pattern(orc, lap, uap, amap)
        cfg = bt.hop_cfg(lap, uap, amap)
        for first, count in ((0, 64), (64 * 12345, 64 * 1000), (_hop.SEQ_LEN - 4096, 4096), (1 << 26, 1 << 22)):
            got = bt.hop_sequence(cfg, first, count)
            assert np.array_equal(got, want[first:first + count]), (hex(lap), first)
        clocks = rng.integers(0, 1 << 32, 50000, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(bt.hop_channels(cfg, clocks), want[clocks & (_hop.SEQ_LEN - 1)])
        orc.orc_piconet_free(pn)
        orc.orc_hop_cache_clear()


def test_argument_checks():
    cfg = bt.hop_cfg(1, 2)
    lib = bt.lib()
    buf = bt.DeviceBuffer(4096)
    assert lib.btbbx_hop_sequence_device(C.byref(cfg), 32, 64, buf.ptr, None) == -3        # unaligned range
    assert lib.btbbx_hop_sequence_device(C.byref(cfg), _hop.SEQ_LEN, 64, buf.ptr, None) == -3
    cfg.afh, cfg.used_channels = 1, 0
    assert lib.btbbx_hop_sequence_device(C.byref(cfg), 0, 64, buf.ptr, None) == -3         # the reference divides by 0 here
    n = C.c_int(0)
    assert not lib.btbbx_hop_reversal_open(C.byref(cfg), 0, 0, 0, C.byref(n))
    buf.free()


@pytest.mark.parametrize("case", HOP["cases"], ids=lambda c: "%06x" % c["lap"])
def test_reversal_matches_reference_fixture(case):
    """btbbx_hop_reversal_* replays the trace btbb_init_hop_reversal / btbb_winnow produced."""
    cfg = bt.hop_cfg(case["lap"], case["uap"], _amap(case))
    obs, trace = case["obs"], case["trace"]
    rev = bt.HopReversal(cfg, case["c0"] & 63, obs[0][1], case["aliased"])
    assert rev.count == trace[0]["n"]
    cand = rev.candidates()
    assert zlib.crc32(cand.astype("<u4").tobytes()) == trace[0]["cand_crc"] and cand[:8].tolist() == trace[0]["cand_head"]
    used = 0                                            # the reference's pn->winnowed
    for k in range(1, len(trace)):
        offs = [o[0] for o in obs[used:k + 1]]
        chans = [o[1] for o in obs[used:k + 1]]
        stop, count, cand0 = rev.winnow(offs, chans)
        used += stop
        assert count == trace[k]["n"] and used == trace[k]["winnowed"], k
        cand = rev.candidates()
        assert zlib.crc32(cand.astype("<u4").tobytes()) == trace[k]["cand_crc"], k
        assert cand[:8].tolist() == trace[k]["cand_head"]
        if count:
            assert cand0 == cand[0]
    assert rev.count == 1 and rev.candidates()[0] == case["c0"]
    rev.close()


def test_reversal_random_against_oracle():
    """Candidate lists after every step, with contradictions, batches of observations, aliasing,
    AFH maps small enough to overflow the reference's own candidate array."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(62))
    for rep, (used, alias) in enumerate(((None, 0), (None, 1), (30, 0), (5, 0), (None, 0), (66, 1))):
        lap, uap = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256))
        amap = _hop.afh_map_bytes(rng, used) if used else None
        pn, seq = _hop.orc_pattern(orc, lap, uap, amap)
        cfg = bt.hop_cfg(lap, uap, amap)
        c0, t0 = int(rng.integers(0, _hop.SEQ_LEN)), int(rng.integers(0, 1 << 27))
        c = pn.contents
        c.first_pkt_time, c.clk_offset, c.aliased = t0, ((c0 & 63) - (t0 & 63)) & 63, alias
        obs = _hop.observations(rng, seq, c0, 30, alias=bool(alias), max_gap=2000)
        if rep == 4:
            obs[3] = (obs[3][0], (obs[3][1] + 7) % 79)          # contradiction -> no candidates
        feed = [1, 1, 3, 1, 2] + [1] * 40                        # observations per btbb_winnow call
        c.pattern_indices[0], c.pattern_channels[0] = obs[0]
        c.packets_observed = 1
        assert orc.orc_init_hop_reversal(alias, pn) > 0
        rev = bt.HopReversal(cfg, c0 & 63, obs[0][1], alias)
        n = c.num_candidates
        assert rev.count == n
        assert np.array_equal(rev.candidates(), np.ctypeslib.as_array(c.clock_candidates, (n,)))
        k = 1
        for step in feed:
            if k >= len(obs) or not (c.flags >> _hop.F_HOP_INIT & 1):
                break
            for idx, ch in obs[k:k + step]:
                c.pattern_indices[c.packets_observed], c.pattern_channels[c.packets_observed] = idx, ch
                c.packets_observed += 1
            k += step
            w0 = c.winnowed
            offs = [c.pattern_indices[i] for i in range(w0, c.packets_observed)]
            chans = [c.pattern_channels[i] for i in range(w0, c.packets_observed)]
            rv = orc.orc_winnow(pn)
            stop, count, cand0 = rev.winnow(offs, chans)
            assert count == rv, (rep, k)
            if rv:
                assert w0 + stop == c.winnowed
                n = c.num_candidates
                assert np.array_equal(rev.candidates(), np.ctypeslib.as_array(c.clock_candidates, (n,))), (rep, k)
            if rv <= 1:
                break
        if rep == 4:
            assert rev.count == 0
        elif used is None or used > 20:
            assert rev.count == 1 and rev.candidates()[0] == c0
        rev.close()
        orc.orc_piconet_free(pn)
        orc.orc_hop_cache_clear()


def test_reversal_empty_list_stops_at_first_observation():
    """A channel the pattern never produces leaves no candidate; the next btbb_winnow of the reference
    then runs channel_winnow once, gets 0 and resets (bluetooth_piconet.c:596-601, 614-620): the batch
    entry reports stop = 0 / count = 0 instead of skipping the (empty) list."""
    orc = _libs.oracle()
    lap, uap = 0x2468AC, 0x51
    pn, _ = _hop.orc_pattern(orc, lap, uap, None)
    c = pn.contents
    c.first_pkt_time, c.clk_offset = 0, 5
    c.pattern_indices[0], c.pattern_channels[0] = 0, 79
    c.packets_observed = 1
    assert orc.orc_init_hop_reversal(0, pn) == 0
    rev = bt.HopReversal(bt.hop_cfg(lap, uap), 5, 79, 0)
    assert rev.count == 0
    c.pattern_indices[1], c.pattern_channels[1] = 40, 3
    c.packets_observed = 2
    assert orc.orc_winnow(pn) == 0 and c.winnowed == 0 and not (c.flags >> _hop.F_HOP_INIT & 1)
    assert rev.winnow([0, 40], [79, 3])[:2] == (0, 0)
    assert rev.winnow([], [])[:2] == (0, 0)                     # nothing to apply: unchanged, no stop
    rev.close()
    orc.orc_piconet_free(pn)
    orc.orc_hop_cache_clear()


def test_process_packet_on_unused_channel_resets(capfd):
    """Packets reported on channel 79 (never hopped on): CLK1-6 is found from the headers, the hop
    reversal opens with an empty candidate list and the first winnow resets the piconet -- state and
    return values equal the oracle's after every packet."""
    from test_gpu_packets import DropIn
    lib, orc = bt.lib(), _libs.oracle()
    rng = np.random.default_rng(_libs.seed(710))
    lap, uap = 0x13579B, 0x6D
    _, seq = _hop.orc_pattern(orc, lap, uap, None)
    pn = C.c_void_p(lib.btbb_piconet_new())
    on = orc.orc_piconet_new()
    lib.btbb_init_piconet(pn, lap)
    orc.orc_init_piconet(on, lap)
    lib.btbb_piconet_set_uap(pn, uap)
    on.contents.UAP = uap
    orc.orc_piconet_set_flag(on, _hop.F_UAP_VALID, 1)
    resets = 0
    for k, (sym, ch, clkn) in enumerate(_hop.piconet_traffic(rng, seq, lap, uap, 123456, 40)):
        d = DropIn(lib, orc, lap)
        d.set_data(sym, 79, clkn)
        a, b = lib.btbb_process_packet(d.p, pn), orc.orc_process_packet(d.o, on)
        assert a == b == 0, k
        d.check(k)
        c = on.contents
        assert _state(lib, pn) == [c.num_candidates, c.winnowed, c.packets_observed, c.total_packets_observed,
                                   c.first_pkt_time, c.flags, c.used_channels], k
        assert lib.btbb_piconet_get_uap(pn) == c.UAP and lib.btbb_piconet_get_clk_offset(pn) == c.clk_offset
        # the whole ladder (CLK1-6 found -> reversal opened empty -> winnow -> reset) runs inside one call
        assert not (c.flags >> _hop.F_HOP_INIT & 1) and not (c.flags >> _hop.F_CLK27_VALID & 1), k
        resets += int(c.total_packets_observed == 0 and c.packets_observed == 0 and c.num_candidates == 0)
        d.close()
    lib.btbb_piconet_unref(pn)
    orc.orc_piconet_free(on)
    orc.orc_hop_cache_clear()
    capfd.readouterr()
    assert resets >= 5


def _state(lib, pn):
    return [int(lib.btbbx_piconet_state(pn, f)) for f in range(7)]


def test_process_packet_to_following(capfd):
    """Drop-in btbb_process_packet with the UAP known in advance: CLK1-6 from headers on the
    GPU, hop reversal on the GPU, CLK1-27 acquisition, FOLLOWING -- every step equals the oracle."""
    from test_gpu_packets import DropIn
    lib, orc = bt.lib(), _libs.oracle()
    followed = 0
    for case in range(3):
        rng = np.random.default_rng(_libs.seed(700 + case))
        lap, uap = int(rng.integers(0, 1 << 24)), int(rng.integers(1, 256))
        _, seq = _hop.orc_pattern(orc, lap, uap, None)
        for rep in range(3):
            c0 = int(rng.integers(0, _hop.SEQ_LEN))
            pn = C.c_void_p(lib.btbb_piconet_new())
            on = orc.orc_piconet_new()
            lib.btbb_init_piconet(pn, lap)
            orc.orc_init_piconet(on, lap)
            lib.btbb_piconet_set_uap(pn, uap)
            on.contents.UAP = uap
            orc.orc_piconet_set_flag(on, _hop.F_UAP_VALID, 1)
            for k, (sym, ch, clkn) in enumerate(_hop.piconet_traffic(rng, seq, lap, uap, c0, 60)):
                d = DropIn(lib, orc, lap)
                d.set_data(sym, ch, clkn)
                a, b = lib.btbb_process_packet(d.p, pn), orc.orc_process_packet(d.o, on)
                assert a == b, (case, rep, k)
                d.check((case, rep, k))
                c = on.contents
                assert _state(lib, pn) == [c.num_candidates, c.winnowed, c.packets_observed, c.total_packets_observed,
                                           c.first_pkt_time, c.flags, c.used_channels], (case, rep, k)
                assert lib.btbb_piconet_get_uap(pn) == c.UAP and lib.btbb_piconet_get_clk_offset(pn) == c.clk_offset
                if c.flags >> _hop.F_HOP_INIT & 1:
                    got = np.zeros(max(c.num_candidates, 1), np.uint32)
                    n = lib.btbbx_piconet_candidates(pn, _libs.ptr(got), len(got))
                    assert n == c.num_candidates
                    assert np.array_equal(got[:n], np.ctypeslib.as_array(c.clock_candidates, (max(n, 1),))[:n])
                d.close()
                if a == -1:
                    followed += 1
                    assert c.clk_offset == -34
                    break
            lib.btbb_piconet_unref(pn)
            orc.orc_piconet_free(on)
        orc.orc_hop_cache_clear()
    capfd.readouterr()
    assert followed >= 6


def test_direct_drop_in_calls(capfd):
    """btbb_piconet_set_afh_map + btbb_init_hop_reversal + btbb_winnow called directly."""
    lib, orc = bt.lib(), _libs.oracle()
    rng = np.random.default_rng(_libs.seed(63))
    lap, uap = 0x777123, 0x3C
    amap = _hop.afh_map_bytes(rng, 64)
    while not amap[0] & 1:                       # channel 0 must be in use: nothing observed yet -> hop on 0
        amap = _hop.afh_map_bytes(rng, 64)
    on, seq = _hop.orc_pattern(orc, lap, uap, amap)
    pn = C.c_void_p(lib.btbb_piconet_new())
    lib.btbb_init_piconet(pn, lap)
    lib.btbb_piconet_set_uap(pn, uap)
    lib.btbb_piconet_set_flag(pn, _hop.F_IS_AFH, 1)
    lib.btbb_piconet_set_afh_map(pn, _libs.ptr(amap))
    assert int(lib.btbbx_piconet_state(pn, 6)) == 64
    # nothing observed yet: pattern_channels[0] == 0, first_pkt_time == 0, clk_offset as set
    lib.btbb_piconet_set_clk_offset(pn, 17)
    on.contents.clk_offset = 17
    a, b = lib.btbb_init_hop_reversal(1, pn), orc.orc_init_hop_reversal(1, on)
    assert a == b and a > 0
    assert int(lib.btbbx_piconet_state(pn, 5)) == on.contents.flags      # IS_ALIASED set, aliasing not applied (H3)
    assert lib.btbb_winnow(pn) == orc.orc_winnow(on) == a                 # no observations: unchanged
    lib.btbb_piconet_unref(pn)
    orc.orc_piconet_free(on)
    orc.orc_hop_cache_clear()
    capfd.readouterr()
