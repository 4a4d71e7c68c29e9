"""Packet-level oracle vs compiled reference: header_present, try_clock, crc_check (all 64
clocks, state carried across trials exactly like btbb_uap_from_header does), decode_header,
decode_payload, btbb_decode, uap_from_header / process_packet sequences."""
import ctypes as C

import numpy as np
import pytest

import _libs
import _pkt
from libbtbb_amd import synth

ref = _libs.ref()
pytestmark = pytest.mark.skipif(ref is None, reason="compiled reference (oracle/_ref) not available")


@pytest.fixture(scope="module")
def orc():
    o = _libs.oracle()
    o.orc_init(2)
    ref.btbb_init(2)
    return o


def test_sixty_four_clock_trials(orc):
    rng = np.random.default_rng(21)
    hist = {}
    for sym, meta in _pkt.random_packets(rng, 240):
        pr = _pkt.Pair(orc, ref, meta["lap"], 0)
        pr.set_data(sym, channel=int(rng.integers(0, 79)), clkn=int(rng.integers(0, 1 << 28)))
        assert orc.orc_header_present(pr.o) == ref.btbb_header_present(pr.r)
        for clock in range(64):
            u1, u2 = orc.orc_try_clock(clock, pr.o), ref.try_clock(clock, pr.r)
            assert u1 == u2
            c1, c2 = orc.orc_crc_check(clock, pr.o), ref.crc_check(clock, pr.r)
            assert c1 == c2, (meta, clock)
            hist[c1] = hist.get(c1, 0) + 1
            if clock % 16 == 5 or c1 > 2:
                pr.check((meta, clock))
        pr.check(meta)
        pr.close()
    assert hist.get(10, 0) > 20 and hist.get(1000, 0) > 3 and hist.get(0, 0) > 0 and hist.get(2, 0) > 0


@pytest.mark.parametrize("fn", ["fhs", "DM", "DH", "EV3", "EV4", "EV5", "HV"])
def test_type_decoders_direct(orc, fn):
    """Each per-type decoder called directly with every packet_type value it can see."""
    rng = np.random.default_rng(hash(fn) % 1000)
    types = {"fhs": [2], "DM": [3, 8, 10, 14, 0], "DH": [4, 9, 11, 15, 0], "EV3": [7], "EV4": [12],
             "EV5": [13], "HV": [5, 6, 7, 1]}[fn]
    for sym, meta in _pkt.random_packets(rng, 60):
        pr = _pkt.Pair(orc, ref, meta["lap"], 0)
        pr.set_data(sym)
        for t in types:
            for clock in (meta["clk6"], (meta["clk6"] + 7) % 64):
                for which in (pr.o.contents, ):
                    which.packet_type = t
                view = _libs.RefPacketView(ref, pr.r.value)
                C.c_uint8.from_address(pr.r.value + view._off("packet_type")).value = t
                pr.o.contents.UAP = meta["uap"]
                C.c_uint8.from_address(pr.r.value + view._off("UAP")).value = meta["uap"]
                r1 = getattr(orc, "orc_" + fn)(clock, pr.o)
                r2 = getattr(ref, fn)(clock, pr.r)
                assert r1 == r2, (fn, t, clock, meta)
                pr.check((fn, t, clock))
        pr.close()


def test_decode_known_clock(orc, capfd):
    rng = np.random.default_rng(22)
    ok = 0
    for sym, meta in _pkt.random_packets(rng, 150, max_sym_errors=1):
        for trial in range(2):
            pr = _pkt.Pair(orc, ref, meta["lap"], 0)
            clkn = (int(rng.integers(0, 1 << 20)) << 7) | (meta["clk6"] << 1) | int(rng.integers(0, 2))
            if trial == 1:
                clkn ^= 2 << int(rng.integers(0, 6))           # wrong clock
            pr.set_data(sym, channel=3, clkn=clkn)
            pr.set_uap(meta["uap"])
            assert orc.orc_decode_header(pr.o) == ref.btbb_decode_header(pr.r) == 0   # CLK6 not valid yet
            pr.set_flag(4, 1)
            h1, h2 = orc.orc_decode_header(pr.o), ref.btbb_decode_header(pr.r)
            assert h1 == h2
            pr.check(("hdr", meta))
            if h1:
                p1, p2 = orc.orc_decode_payload(pr.o), ref.btbb_decode_payload(pr.r)
                assert p1 == p2
                ok += p1 in (10, 1000)
                pr.check(("payload", meta))
                assert orc.orc_packet_header_packed(pr.o) == ref.btbb_packet_get_header_packed(pr.r)
                b1, b2 = np.zeros(400, np.uint8), np.zeros(400, np.uint8)
                n1, n2 = orc.orc_payload_packed(pr.o, _libs.ptr(b1)), ref.btbb_get_payload_packed(pr.r, _libs.ptr(b2))
                assert n1 == n2 and (b1 == b2).all()
            d1, d2 = orc.orc_decode(pr.o), ref.btbb_decode(pr.r)
            assert d1 == d2
            pr.check(("decode", meta))
            pr.close()
    capfd.readouterr()     # swallow the reference's printf output
    assert ok > 25


def _pn_pair(orc, lap):
    o = orc.orc_piconet_new()
    r = C.c_void_p(ref.btbb_piconet_new())
    orc.orc_init_piconet(o, lap)
    ref.btbb_init_piconet(r, lap)
    return o, r


def _pn_check(orc, o, r):
    cand = (C.c_int * 64)()
    ref.refint_piconet_candidates(r, cand)
    assert list(cand) == list(o.contents.clock6_candidates)
    assert o.contents.flags == ref.refint_piconet_flags(r)
    assert o.contents.UAP == ref.btbb_piconet_get_uap(r)
    assert o.contents.clk_offset == ref.btbb_piconet_get_clk_offset(r)
    assert o.contents.packets_observed == ref.refint_piconet_packets_observed(r)
    assert o.contents.total_packets_observed == ref.refint_piconet_total_packets_observed(r)
    assert o.contents.first_pkt_time == ref.refint_piconet_first_pkt_time(r)


def test_uap_from_header_sequences(orc, capfd):
    """Several packets of one piconet fed in time order: candidate elimination must agree."""
    rng = np.random.default_rng(23)
    found = 0
    for seq in range(25):
        lap, uap = int(rng.integers(0, 1 << 24)), int(rng.integers(1, 256))
        o, r = _pn_pair(orc, lap)
        clk = int(rng.integers(0, 1 << 26))
        for k in range(12):
            clk += int(rng.integers(1, 40)) * 2
            clk6 = (clk >> 1) & 0x3F
            # types without CRC first (inconclusive), then a DM1/DH1 to clinch it
            t = [0, 1, 9, 6][k % 4] if k < int(rng.integers(2, 9)) else [3, 4, 10, 2][k % 4]
            body = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
            sym = synth.build_packet(lap, uap, clk6, t, lt_addr=1, body=body,
                                     fhs_bits=synth.fhs_payload(lap, uap, 1, 2, rng))
            sym = np.concatenate([sym, rng.integers(0, 2, 50, dtype=np.uint8)])
            if rng.random() < 0.3:
                sym[int(rng.integers(68, len(sym)))] ^= 1
            # local clock differs from the piconet clock by a constant offset
            clkn = (clk + 2 * 17) & 0xFFFFFFF
            pr = _pkt.Pair(orc, ref, lap, 0)
            pr.set_data(sym, channel=int(rng.integers(0, 79)), clkn=clkn)
            if seq % 2:
                a, b = orc.orc_uap_from_header(pr.o, o), ref.btbb_uap_from_header(pr.r, r)
            else:
                a, b = orc.orc_process_packet(pr.o, o), ref.btbb_process_packet(pr.r, r)
            assert a == b
            pr.check((seq, k))
            _pn_check(orc, o, r)
            pr.close()
            if orc.orc_piconet_get_flag(o, 2) and orc.orc_piconet_get_flag(o, 4):
                found += o.contents.UAP == uap
                if seq % 2 == 0:
                    break      # process_packet would now enter hop reversal (out of scope)
        orc.orc_piconet_free(o)
        ref.btbb_piconet_unref(r)
    capfd.readouterr()
    assert found >= 15
