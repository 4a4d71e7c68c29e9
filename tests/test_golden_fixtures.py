"""Golden fixtures generated from the compiled reference (tests/golden/make_golden.py):
the oracle must reproduce them everywhere (CPU), the HIP path on the GPU (`-m gpu`)."""
import json
import os

import numpy as np
import pytest

import _libs
from libbtbb_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SCAN = json.load(open(os.path.join(HERE, "golden", "scan_hits.json")))
PK = np.load(os.path.join(HERE, "golden", "packets.npz"))


def _case_stream(case):
    words, _ = synth.make_stream(case["seed"], case["nwords"], stride=case["stride"], lap=case["lap"])
    return words, np.ascontiguousarray(synth.unpack_bits(words))


@pytest.mark.parametrize("case", SCAN["cases"], ids=lambda c: c["name"])
def test_oracle_scan_matches_reference_fixture(case):
    orc = _libs.oracle()
    orc.orc_reset_syndrome_map()
    orc.orc_init(SCAN["init_max_ac_errors"])
    _, sym = _case_stream(case)
    lap = _libs.LAP_ANY if case["lap"] is None else case["lap"]
    total = 0
    for me, want in case["hits"].items():
        got = _libs.orc_find_all(sym, case["search_bits"], lap, int(me))
        assert got == [tuple(h) for h in want], (case["name"], me)
        total += len(got)
    assert total > 100


def test_oracle_packets_match_reference_fixture():
    orc = _libs.oracle()
    orc.orc_init(2)
    import _pkt
    n = len(PK["lengths"])
    payload = np.unpackbits(PK["payload"], axis=1, bitorder="little")[:, :2744]
    for i in range(n):
        sym = np.ascontiguousarray(synth.unpack_bits(PK["words"][i], int(PK["lengths"][i])))
        lap, uap, clk6, _ = (int(x) for x in PK["meta"][i])
        p = orc.orc_packet_new()
        orc.orc_packet_init_found(p, lap, 0)
        orc.orc_packet_set_data(p, _libs.ptr(sym), len(sym), 0, 0)
        for clock in range(64):
            u = orc.orc_try_clock(clock, p)
            rv = orc.orc_crc_check(clock, p)
            assert (u, p.contents.packet_type, rv) == tuple(int(x) for x in PK["trials"][i, clock]), (i, clock)
        orc.orc_packet_free(p)
        p = orc.orc_packet_new()
        orc.orc_packet_init_found(p, lap, 0)
        orc.orc_packet_set_data(p, _libs.ptr(sym), len(sym), 0, clk6 << 1)
        p.contents.UAP = uap
        orc.orc_packet_set_flag(p, 2, 1)
        orc.orc_packet_set_flag(p, 4, 1)
        present = orc.orc_header_present(p)
        h = orc.orc_decode_header(p)
        r = orc.orc_decode_payload(p) if h else 0
        st = _pkt.orc_state(p)
        got = (present, h, r, st["packet_type"], st["packet_lt_addr"], st["packet_flags"], st["packet_hec"],
               st["payload_length"], st["payload_header_length"],
               int(sum(int(b) << k for k, b in enumerate(st["packet_header"]))))
        assert got == tuple(int(x) for x in PK["decode"][i]), i
        assert (st["payload"] == payload[i]).all(), i
        assert st["flags"] == int(PK["flags"][i]), i
        orc.orc_packet_free(p)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SCAN["cases"], ids=lambda c: c["name"])
def test_gpu_scan_matches_reference_fixture(case):
    import libbtbb_amd as bt
    bt.lib().btbbx_shutdown()
    bt.init(SCAN["init_max_ac_errors"])
    words, _ = _case_stream(case)
    lap = bt.LAP_ANY if case["lap"] is None else case["lap"]
    for me, want in case["hits"].items():
        hits = bt.scan_words(words, case["search_bits"], lap, int(me))
        got = [(int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in hits]
        assert got == [tuple(h) for h in want], (case["name"], me)


@pytest.mark.gpu
def test_gpu_packets_match_reference_fixture():
    import libbtbb_amd as bt
    bt.init(2)
    n = len(PK["lengths"])
    words = np.ascontiguousarray(PK["words"])
    pin = np.zeros(n, bt.PKTIN_DTYPE)
    pin["length"] = PK["lengths"]
    pin["flags"] = 1
    trials = bt.run_trials(words, pin)
    got = np.stack([trials["uap"], trials["type"], trials["rv"]], axis=-1).astype(np.int32)
    assert (got == PK["trials"]).all(), np.argwhere(got != PK["trials"])[:5]
    pin["clkn"] = PK["meta"][:, 2]
    pin["uap"] = PK["meta"][:, 1]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    out = bt.run_decode(words, pin)
    payload = np.unpackbits(PK["payload"], axis=1, bitorder="little")[:, :2744]
    for i in range(n):
        o, d = out[i], PK["decode"][i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == tuple(int(x) for x in d[:3]), i
        assert int(o["header_packed"]) == int(d[9]) and int(o["flags"]) == int(PK["flags"][i]), i
        if d[1]:
            assert (int(o["type"]), int(o["lt_addr"]), int(o["hdr_flags"]), int(o["hec"]),
                    int(o["payload_length"]), int(o["payload_header_length"])) == tuple(int(x) for x in d[3:9]), i
            assert (synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744) == payload[i]).all(), i


HOP = json.load(open(os.path.join(HERE, "golden", "hop.json")))


@pytest.mark.parametrize("case", HOP["cases"], ids=lambda c: "%06x" % c["lap"])
def test_oracle_hop_matches_reference_fixture(case):
    """Whole hop pattern digests and a CLK1-27 reversal trace recorded from the reference."""
    import ctypes as C
    import hashlib
    import zlib
    import _hop
    orc = _libs.oracle()
    amap = None if case["afh_map"] is None else np.array(case["afh_map"], np.uint8)
    pn, seq = _hop.orc_pattern(orc, case["lap"], case["uap"], amap)
    assert seq[:512].tolist() == case["head"]
    assert [zlib.crc32(seq[i << 20:(i + 1) << 20].tobytes()) for i in range(128)] == case["crc32_per_mib"]
    assert hashlib.sha256(seq.tobytes()).hexdigest() == case["sha256"]
    c = pn.contents
    c.first_pkt_time = case["t0"]
    c.clk_offset = ((case["c0"] & 63) - (case["t0"] & 63)) & 63
    c.aliased = case["aliased"]
    for k, ((idx, ch), want) in enumerate(zip(case["obs"], case["trace"])):
        c.pattern_indices[c.packets_observed] = idx
        c.pattern_channels[c.packets_observed] = ch
        c.packets_observed += 1
        c.total_packets_observed += 1
        rv = orc.orc_init_hop_reversal(case["aliased"], pn) if k == 0 else orc.orc_winnow(pn)
        assert (rv, c.num_candidates, c.winnowed, c.flags, c.clk_offset) == \
            (want["rv"], want["n"], want["winnowed"], want["flags"], want["clk_offset"]), k
        if c.flags >> _hop.F_HOP_INIT & 1:
            cand = np.array([c.clock_candidates[i] for i in range(c.num_candidates)], "<u4")
            assert zlib.crc32(cand.tobytes()) == want["cand_crc"]
            assert cand[:8].tolist() == want["cand_head"]
    assert case["trace"][-1]["n"] == 1 and c.clock_candidates[0] == case["c0"]
    orc.orc_piconet_free(pn)
    orc.orc_hop_cache_clear()
