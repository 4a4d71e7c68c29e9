"""Hop-sequence generation and CLK1-27 reversal: oracle (oracle/btbb_oracle_hop.c) against the
compiled reference (lib/src/bluetooth_piconet.c:170-645) -- permutation, whole 2^27-entry
sequences with and without AFH, single_hop, the pattern-cache quirk, candidate lists after
every winnowing step including resets, aliasing and the AFH heuristics."""
import ctypes as C

import numpy as np
import pytest

import _hop
import _libs

ref = _libs.ref()
pytestmark = pytest.mark.skipif(ref is None, reason="compiled reference (oracle/_ref) not available")


@pytest.fixture(scope="module")
def orc():
    o = _libs.oracle()
    yield o
    o.orc_hop_cache_clear()


def _ref_pattern(lap, uap, afh_map=None):
    r = C.c_void_p(ref.btbb_piconet_new())
    ref.btbb_init_piconet(r, lap)
    ref.btbb_piconet_set_uap(r, uap)
    if afh_map is not None:
        ref.btbb_piconet_set_flag(r, _hop.F_IS_AFH, 1)
        ref.btbb_piconet_set_afh_map(r, _libs.ptr(afh_map))
    else:
        ref.btbb_piconet_set_channel_seen(r, 0)     # H6, see _hop.orc_pattern
        ref.get_hop_pattern(r)
    return r, _hop.seq_view(ref.refint_piconet_sequence(r))


def _params(r):
    buf = (C.c_int * 84)()
    ref.refint_piconet_hop_params(r, buf)
    return list(buf)


def _orc_params(o):
    c = o.contents
    return [c.a1, c.b, c.c1, c.d1, c.e] + list(c.bank)


def test_perm5_exhaustive(orc):
    for z in range(32):
        for ph in range(32):
            for pl in range(0, 512, 1 if z % 8 == 0 else 7):
                assert orc.orc_perm5(z, ph, pl) == ref.perm5(z, ph, pl)


# one module-wide set of patterns: each costs 128 MiB per side
CASES = [(0x9E8B33, 0x00, None), (0x123456, 0xA7, None), (0xC0FFEE, 0x5B, 61), (0x00F00D, 0xFF, 50)]


@pytest.fixture(scope="module")
def patterns(orc):
    rng = np.random.default_rng(5)
    out = []
    for lap, uap, used in CASES:
        amap = _hop.afh_map_bytes(rng, used) if used else None
        o, so = _hop.orc_pattern(orc, lap, uap, amap)
        r, sr = _ref_pattern(lap, uap, amap)
        out.append((lap, uap, amap, o, so, r, sr))
    return out


def test_whole_sequences(patterns, capfd):
    for lap, uap, amap, o, so, r, sr in patterns:
        assert _orc_params(o) == _params(r), hex(lap)
        assert np.array_equal(so, sr), hex(lap)
        assert so.max() < 79
        if amap is not None:
            allowed = {c for c in range(79) if amap[c // 8] >> (c % 8) & 1}
            assert set(np.unique(so[: 1 << 22]).tolist()) <= allowed
    capfd.readouterr()


def test_single_hop(orc, patterns):
    rng = np.random.default_rng(6)
    for lap, uap, amap, o, so, r, sr in patterns:
        for clock in rng.integers(0, 1 << 28, 3000).tolist():
            a, b = orc.orc_single_hop(clock, o), ref.single_hop(clock, r)
            assert a == b
            if amap is None:      # H2: gen_hops and single_hop only agree without AFH
                assert ord(a) == so[clock >> 1]


def test_cache_is_keyed_by_address_only(orc, patterns, capfd):
    """H1: a second piconet with the same address but AFH on gets the cached non-AFH pattern."""
    lap, uap = CASES[0][0], CASES[0][1]
    amap = _hop.afh_map_bytes(np.random.default_rng(9), 50)
    o, so = _hop.orc_pattern(orc, lap, uap, amap)
    r, sr = _ref_pattern(lap, uap, amap)
    assert ref.refint_piconet_sequence(r) == ref.refint_piconet_sequence(patterns[0][5])
    assert o.contents.sequence == patterns[0][3].contents.sequence
    assert np.array_equal(so[: 1 << 20], sr[: 1 << 20])
    capfd.readouterr()


def _reversal_pair(orc, lap, uap, amap, first_pkt_time, c0, alias):
    """Oracle and reference piconets prepared as btbb_uap_from_header leaves them when the UAP
    and CLK1-6 have just been found (UAP/CLK6 valid, clk_offset so that CLK1-6 == c0 & 63)."""
    o, so = _hop.orc_pattern(orc, lap, uap, amap)
    r, sr = _ref_pattern(lap, uap, amap)
    clk_offset = ((c0 & 63) - (first_pkt_time & 63)) & 63
    o.contents.first_pkt_time = first_pkt_time
    o.contents.clk_offset = clk_offset
    o.contents.aliased = alias
    ref.refint_piconet_set_first_pkt_time(r, first_pkt_time)
    ref.btbb_piconet_set_clk_offset(r, clk_offset)
    ref.refint_piconet_set_aliased(r, alias)
    for pn_set in (lambda f: orc.orc_piconet_set_flag(o, f, 1), lambda f: ref.btbb_piconet_set_flag(r, f, 1)):
        pn_set(_hop.F_CLK6_VALID)
        pn_set(_hop.F_GOT_FIRST)
    return o, so, r


def _observe(o, r, index, channel):
    c = o.contents
    c.pattern_indices[c.packets_observed] = index
    c.pattern_channels[c.packets_observed] = channel
    c.packets_observed += 1
    c.total_packets_observed += 1
    ref.refint_piconet_observe(r, index, channel)


def _same_state(o, r, tag):
    c = o.contents
    assert c.flags == ref.refint_piconet_flags(r), tag
    assert c.num_candidates == ref.refint_piconet_num_candidates(r), tag
    assert c.winnowed == ref.refint_piconet_winnowed(r), tag
    assert c.packets_observed == ref.refint_piconet_packets_observed(r), tag
    assert c.clk_offset == ref.btbb_piconet_get_clk_offset(r), tag
    if c.flags >> _hop.F_HOP_INIT & 1:
        n = c.num_candidates
        rc = ref.refint_piconet_clock_candidates(r)
        assert [c.clock_candidates[i] for i in range(n)] == [rc[i] for i in range(n)], tag


@pytest.mark.parametrize("case,alias", [(0, 0), (1, 0), (2, 0), (3, 0), (1, 1)])
def test_reversal_finds_the_clock(orc, patterns, case, alias, capfd):
    lap, uap, amap = patterns[case][:3]
    rng = np.random.default_rng(100 + case + alias)
    for rep in range(3):
        c0 = int(rng.integers(0, _hop.SEQ_LEN))
        t0 = int(rng.integers(0, 1 << 27))
        o, so, r = _reversal_pair(orc, lap, uap, amap, t0, c0, alias)
        obs = _hop.observations(rng, so, c0, 40, alias=bool(alias))
        _observe(o, r, *obs[0])
        a, b = orc.orc_init_hop_reversal(alias, o), ref.btbb_init_hop_reversal(alias, r)
        assert a == b and a > 1000
        _same_state(o, r, (case, rep, "init"))
        assert orc.orc_winnow(o) == ref.btbb_winnow(r)
        _same_state(o, r, (case, rep, "winnow0"))
        done = False
        for k, (idx, ch) in enumerate(obs[1:]):
            _observe(o, r, idx, ch)
            a, b = orc.orc_winnow(o), ref.btbb_winnow(r)
            assert a == b
            _same_state(o, r, (case, rep, k))
            if a == 1:
                assert o.contents.flags >> _hop.F_CLK27_VALID & 1
                assert o.contents.clock_candidates[0] == c0
                assert o.contents.clk_offset == C.c_int32(((c0 << 1) - (t0 << 1)) & 0xFFFFFFFF).value
                done = True
                break
        assert done or amap is not None
        # H5: the deciding packet is applied again by a further call
        assert orc.orc_winnow(o) == ref.btbb_winnow(r)
        _same_state(o, r, (case, rep, "again"))
    capfd.readouterr()


def test_reversal_reset_on_contradiction(orc, patterns, capfd):
    lap, uap, amap = patterns[1][:3]
    rng = np.random.default_rng(77)
    c0, t0 = 0x2345678, 0x1111
    o, so, r = _reversal_pair(orc, lap, uap, amap, t0, c0, 0)
    obs = _hop.observations(rng, so, c0, 6)
    _observe(o, r, *obs[0])
    assert orc.orc_init_hop_reversal(0, o) == ref.btbb_init_hop_reversal(0, r)
    # several observations at once, the third one contradicts every candidate
    _observe(o, r, *obs[1])
    _observe(o, r, obs[2][0], (obs[2][1] + 1) % 79)
    _observe(o, r, *obs[3])
    rv = orc.orc_winnow(o)
    assert rv == ref.btbb_winnow(r)
    _same_state(o, r, "contradiction")
    if rv == 0:
        assert not (o.contents.flags >> _hop.F_HOP_INIT & 1) and o.contents.packets_observed == 0
    assert orc.orc_winnow(o) == ref.btbb_winnow(r)
    _same_state(o, r, "after reset")
    capfd.readouterr()


def test_afh_heuristics(orc, patterns, capfd):
    """H4 (below-array reads at winnowed == 0) and the consecutive-slot rule."""
    lap, uap, amap = patterns[0][:3]
    so = patterns[0][4]
    # a clock whose hop is channel 0, so that the top byte of pattern_indices[999] (0) matches
    c0 = int(np.flatnonzero(so[: 1 << 16] == 0)[3])
    for cand63, expect in ((-1, 1), (0x47, 0)):
        o, _, r = _reversal_pair(orc, lap, uap, amap, 5, c0, 0)
        o.contents.clock6_candidates[63] = cand63
        ref.refint_piconet_set_candidate6(r, 63, cand63)
        _observe(o, r, 0, 0)
        assert orc.orc_init_hop_reversal(0, o) == ref.btbb_init_hop_reversal(0, r)
        assert orc.orc_winnow(o) == ref.btbb_winnow(r)
        _same_state(o, r, ("H4", cand63))
        assert (o.contents.flags >> _hop.F_LOOKS_AFH & 1) == expect
    # two consecutive slots on one channel -> LOOKS_LIKE_AFH
    lap, uap, amap = patterns[2][:3]
    so = patterns[2][4]
    pairs = np.flatnonzero(so[1000:2000000] == so[1001:2000001]) + 1000
    c0 = int(pairs[0])
    o, _, r = _reversal_pair(orc, lap, uap, amap, 9, c0, 0)
    _observe(o, r, 0, int(so[c0]))
    _observe(o, r, 1, int(so[c0 + 1]))
    assert orc.orc_init_hop_reversal(0, o) == ref.btbb_init_hop_reversal(0, r)
    assert orc.orc_winnow(o) == ref.btbb_winnow(r)
    _same_state(o, r, "consecutive")
    assert o.contents.flags >> _hop.F_LOOKS_AFH & 1
    capfd.readouterr()


def test_process_packet_to_following(orc, patterns, capfd):
    """btbb_process_packet end to end with the UAP known in advance: CLK1-6 from headers, hop
    reversal, CLK1-27 acquisition, then FOLLOWING (bluetooth_piconet.c:501-543, 851-899)."""
    import _pkt
    orc.orc_init(2)
    ref.btbb_init(2)
    followed = 0
    for case in (1, 0):
        lap, uap, amap, _, so = patterns[case][:5]
        if uap == 0:
            continue          # "have UAP" is tested as UAP != 0 (:882)
        rng = np.random.default_rng(500 + case)
        for rep in range(4):
            c0 = int(rng.integers(0, _hop.SEQ_LEN))
            o = orc.orc_piconet_new()
            r = C.c_void_p(ref.btbb_piconet_new())
            orc.orc_init_piconet(o, lap)
            ref.btbb_init_piconet(r, lap)
            o.contents.UAP = uap
            orc.orc_piconet_set_flag(o, _hop.F_UAP_VALID, 1)
            ref.btbb_piconet_set_uap(r, uap)
            for k, (sym, ch, clkn) in enumerate(_hop.piconet_traffic(rng, so, lap, uap, c0, 60)):
                pr = _pkt.Pair(orc, ref, lap, 0)
                pr.set_data(sym, channel=ch, clkn=clkn)
                a, b = orc.orc_process_packet(pr.o, o), ref.btbb_process_packet(pr.r, r)
                assert a == b, (case, rep, k)
                pr.check((case, rep, k))
                _same_state(o, r, (case, rep, k))
                assert o.contents.UAP == ref.btbb_piconet_get_uap(r)
                pr.close()
                if a == -1:
                    followed += 1
                    assert o.contents.flags >> _hop.F_FOLLOWING & 1
                    # master clock = local clock - 34 half slots
                    assert o.contents.clk_offset == -34
                    break
    capfd.readouterr()
    assert followed >= 3
