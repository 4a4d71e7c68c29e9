"""GPU parity for the bit chain behind an access code: 64-clock trial tables, header /
payload decode (batch API) and the drop-in packet functions, all against the oracle."""
import ctypes as C

import numpy as np
import pytest

import _libs
import _pkt
import libbtbb_amd as bt
from libbtbb_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def ready():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    bt.init(2)
    _libs.oracle().orc_init(2)


def _oracle_trials(orc, sym, entry_type=0, entry_uap=0, whitened=1):
    p = orc.orc_packet_new()
    orc.orc_packet_init_found(p, 0, 0)
    orc.orc_packet_set_flag(p, 0, whitened)
    orc.orc_packet_set_data(p, _libs.ptr(sym), len(sym), 0, 0)
    p.contents.packet_type = entry_type
    p.contents.UAP = entry_uap
    out = []
    for clock in range(64):
        # every trial starts from the entry state: only type/UAP can leak between trials, and
        # only when FEC 1/3 fails, which does not depend on the clock
        p.contents.packet_type = entry_type
        p.contents.UAP = entry_uap
        u = orc.orc_try_clock(clock, p)
        rv = orc.orc_crc_check(clock, p)
        out.append((u, p.contents.packet_type, rv))
    orc.orc_packet_free(p)
    return out


def test_trial_tables():
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(31))
    pk = _pkt.random_packets(rng, 360)
    syms = [np.ascontiguousarray(s[:bt.MAX_SYMBOLS]) for s, _ in pk]
    words, lengths = bt.packets_to_words(syms)
    pin = np.zeros(len(syms), bt.PKTIN_DTYPE)
    pin["length"] = lengths
    pin["flags"] = 1
    pin["type"] = rng.integers(0, 16, len(syms))
    pin["uap"] = rng.integers(0, 256, len(syms))
    pin["flags"][::17] = 0                      # a few unwhitened
    got = bt.run_trials(words, pin)
    # small batches take the one-workgroup-per-trial kernel: same table
    for k in (1, 40, 128):
        assert np.array_equal(bt.run_trials(words[:k], pin[:k]), got[:k]), k
    hist = {}
    for i, s in enumerate(syms):
        want = _oracle_trials(orc, s, int(pin["type"][i]), int(pin["uap"][i]), int(pin["flags"][i]) & 1)
        g = [(int(t["uap"]), int(t["type"]), int(t["rv"])) for t in got[i]]
        assert g == want, (i, pk[i][1], [j for j in range(64) if g[j] != want[j]][:5])
        for _, _, rv in want:
            hist[rv] = hist.get(rv, 0) + 1
    assert hist.get(10, 0) > 30 and hist.get(1000, 0) > 5 and hist.get(0, 0) > 0 and hist.get(2, 0) > 0


def test_trial_tables_large_batch():
    """More packets than one workgroup batch holds, every trial against the oracle: the kernel for large
    batches works the FEC 2/3 and CRC prefix of a packet out once and evaluates a trial from tables, so
    truncated captures, junk behind short packets and unwhitened packets all take their own paths."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(47))
    pk = _pkt.random_packets(rng, 1500, max_sym_errors=4)
    syms = [np.ascontiguousarray(s[:bt.MAX_SYMBOLS]) for s, _ in pk]
    words, lengths = bt.packets_to_words(syms)
    pin = np.zeros(len(syms), bt.PKTIN_DTYPE)
    pin["length"] = lengths
    pin["flags"] = 1
    pin["type"] = rng.integers(0, 16, len(syms))
    pin["uap"] = rng.integers(0, 256, len(syms))
    pin["flags"][::11] = 0
    got = bt.run_trials(words, pin)
    bad = []
    for i, s in enumerate(syms):
        want = _oracle_trials(orc, s, int(pin["type"][i]), int(pin["uap"][i]), int(pin["flags"][i]) & 1)
        g = [(int(t["uap"]), int(t["type"]), int(t["rv"])) for t in got[i]]
        if g != want:
            bad.append((i, pk[i][1], len(s), [(j, g[j], want[j]) for j in range(64) if g[j] != want[j]][:3]))
    assert not bad, bad[:5]


def _oracle_decode(orc, sym, clkn, uap, clk_valid=True):
    p = orc.orc_packet_new()
    orc.orc_packet_init_found(p, 0, 0)
    orc.orc_packet_set_data(p, _libs.ptr(sym), len(sym), 0, clkn << 1)
    p.contents.UAP = uap
    orc.orc_packet_set_flag(p, 2, 1)
    orc.orc_packet_set_flag(p, 4, 1 if clk_valid else 0)
    present = orc.orc_header_present(p)
    h = orc.orc_decode_header(p)
    r = orc.orc_decode_payload(p) if h else 0
    st = _pkt.orc_state(p)
    orc.orc_packet_free(p)
    return present, h, r, st


def _ref_decode(ref, sym, clkn, uap, clk_valid=True):
    """The same calls on the compiled, unmodified reference (oracle/_ref): btbb_packet_set_data, btbb_header_present,
    btbb_decode_header, btbb_decode_payload (lib/src/bluetooth_packet.c:467-480, 1198-1297, 1371-1408); the packet's fields
    are read through their real offsets."""
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    p = C.c_void_p(ref.btbb_packet_new())
    view = _libs.RefPacketView(ref, p.value)
    C.c_uint32.from_address(p.value + view._off("LAP")).value = 0
    C.c_uint8.from_address(p.value + view._off("ac_errors")).value = 0
    C.c_uint32.from_address(p.value + view._off("flags")).value = 0
    ref.btbb_packet_set_flag(p, 0, 1)
    ref.btbb_packet_set_data(p, _libs.ptr(sym), len(sym), 0, clkn << 1)
    ref.btbb_packet_set_uap(p, uap)
    ref.btbb_packet_set_flag(p, 4, 1 if clk_valid else 0)
    present = ref.btbb_header_present(p)
    h = ref.btbb_decode_header(p)
    r = ref.btbb_decode_payload(p) if h else 0
    st = _pkt.ref_state(ref, p)
    ref.btbb_packet_unref(p)
    return present, h, r, st


def _both_decode(orc, sym, clkn, uap, ctx=None):
    """The oracle port's verdict and state -- and, where the compiled reference is at hand (it is on the GPU box: oracle/_ref
    travels with the snapshot), the same from it, which must agree in every field the decoders write."""
    got = _oracle_decode(orc, sym, clkn, uap)
    ref = _libs.ref()
    if ref is not None:
        want = _ref_decode(ref, sym, clkn, uap)
        assert got[:3] == want[:3], (ctx, got[:3], want[:3])
        _pkt.assert_same(got[3], want[3], ctx)
        _both_decode.with_ref += 1
    return got


_both_decode.with_ref = 0


def test_batch_decode():
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(32))
    pk = _pkt.random_packets(rng, 300, max_sym_errors=2)
    syms = [np.ascontiguousarray(s[:bt.MAX_SYMBOLS]) for s, _ in pk]
    words, lengths = bt.packets_to_words(syms)
    pin = np.zeros(len(syms), bt.PKTIN_DTYPE)
    pin["length"] = lengths
    clk = np.array([m["clk6"] for _, m in pk], dtype=np.uint32)
    wrong = rng.random(len(syms)) < 0.2
    clk[wrong] ^= 5
    pin["clkn"] = clk | (rng.integers(0, 1 << 20, len(syms)).astype(np.uint32) << 6)
    pin["uap"] = [m["uap"] for _, m in pk]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    pin["flags"][::23] &= ~np.uint32(1 << 4)                  # CLK6 unknown -> header fails
    out = bt.run_decode(words, pin)
    good = 0
    for i, s in enumerate(syms):
        present, h, r, st = _oracle_decode(orc, s, int(pin["clkn"][i]), int(pin["uap"][i]),
                                          bool(pin["flags"][i] & 16))
        o = out[i]
        ctx = (i, pk[i][1])
        assert int(o["header_present"]) == present, ctx
        assert int(o["header_rv"]) == h and int(o["payload_rv"]) == r, ctx
        assert int(o["flags"]) == st["flags"], ctx
        assert int(o["header_packed"]) == int(sum(int(b) << k for k, b in enumerate(st["packet_header"]))), ctx
        if h:
            assert (int(o["type"]), int(o["lt_addr"]), int(o["hdr_flags"]), int(o["hec"])) == \
                   (st["packet_type"], st["packet_lt_addr"], st["packet_flags"], st["packet_hec"]), ctx
            assert int(o["payload_length"]) == st["payload_length"], ctx
            assert int(o["payload_header_length"]) == st["payload_header_length"], ctx
            assert (int(o["llid"]), int(o["flow"])) == (st["payload_llid"], st["payload_flow"]), ctx
            assert int(o["payload_header"]) == int(sum(int(b) << k for k, b in enumerate(st["payload_header"]))), ctx
            bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
            assert (bits == st["payload"]).all(), (ctx, np.nonzero(bits != st["payload"])[0][:8])
            good += r in (10, 1000)
    assert good > 30


def test_decode_from_the_streams_many_workgroups_dirty_records():
    """The same identity at a size where decode_hits_kernel's workgroups are full and sort their packets by decoder
    and length (2 600 hits, every type, lengths mixed), with enough symbol errors that FEC 2/3 blocks fail in the
    middle of DM payloads (the reference then writes no payload at all), and with records that hold random bytes on
    entry: whatever the decoders do not assign must still be there afterwards."""
    rng = np.random.default_rng(_libs.seed(59))
    n_streams, n_words = 5, 1 << 15
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    pk = _pkt.random_packets(rng, 2600, max_sym_errors=9)
    rows = []
    pos = [64 + int(rng.integers(0, 64)) for _ in range(n_streams)]
    for i, (s, meta) in enumerate(pk):
        st = i % n_streams
        s = s[:bt.MAX_SYMBOLS]
        if pos[st] + len(s) + 200 > n_words * 64:
            continue
        sym[st, pos[st]:pos[st] + len(s)] = s
        rows.append((st, pos[st], meta))
        pos[st] += len(s) + int(rng.integers(1, 90))
    assert len(rows) > 1500
    hits = np.zeros(len(rows), bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(len(rows), bt.PKTIN_DTYPE)
    pin["clkn"] = [r[2]["clk6"] for r in rows]
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    dirty = np.frombuffer(rng.integers(0, 256, len(rows) * bt.PKTOUT_DTYPE.itemsize, dtype=np.uint8).tobytes(),
                          dtype=bt.PKTOUT_DTYPE).copy()
    dirty["payload_length"] &= 0x1ff
    dirty["payload_header_length"] &= 3
    direct, len_d = bt.run_decode_hits(words, hits, pin, init_out=dirty)
    two_step, len_g = bt.run_decode_hits(words, hits, pin, via_gather=True, init_out=dirty)
    assert np.array_equal(len_d, len_g)
    bad = [i for i in range(len(rows)) if direct[i].tobytes() != two_step[i].tobytes()]
    assert not bad, (len(bad), bad[:5], [rows[i][2].get("type") for i in bad[:5]])
    rv = direct["payload_rv"]
    assert (rv == 10).sum() > 100 and (rv == 0).sum() > 50 and (direct["header_rv"] == 1).sum() > 500


LONG_TYPES = (synth.TYPE_DM3, synth.TYPE_DH3, synth.TYPE_DM5, synth.TYPE_DH5)
LONG_MAXBODY = {synth.TYPE_DM3: 121, synth.TYPE_DH3: 183, synth.TYPE_DM5: 224, synth.TYPE_DH5: 339,
                synth.TYPE_DM1: 17, synth.TYPE_DH1: 27, synth.TYPE_EV4: 120, synth.TYPE_EV5: 180}


def _long_capture(rng, n_packets, n_streams, n_words, others=(synth.TYPE_DM1, synth.TYPE_DH1, synth.TYPE_FHS, synth.TYPE_EV4, synth.TYPE_EV5),
                  longs=None):
    """Streams of noise with mostly multi-slot DM / DH packets in them: full-length bodies, bodies around the 256-bit
    boundary where the wave phase of decode_hits_kernel takes over, every payload_length mod 8, symbol errors in the
    payload (FEC 2/3 blocks that fail), captures cut short; -> (symbols, rows of (stream, offset, meta))."""
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    rows = []
    pos = [64 + int(rng.integers(0, 64)) for _ in range(n_streams)]
    for i in range(n_packets):
        st = i % n_streams
        t = (longs or LONG_TYPES)[(i // n_streams) % 4] if i % 5 else others[(i // 5) % len(others)]
        mb = LONG_MAXBODY.get(t, 0)
        k = i % 7
        if k == 0:
            nb = mb
        elif k == 1 and t in LONG_TYPES:
            nb = int(rng.integers(24, 40))                    # payload_length around 32 bytes = 256 bits
        elif k == 2:
            nb = max(0, mb - int(rng.integers(0, 9)))         # the last word of the payload at every fill
        else:
            nb = int(rng.integers(0, mb + 1))
        lap, uap, clk6 = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 64))
        p = synth.build_packet(lap, uap, clk6, t, lt_addr=int(rng.integers(0, 8)), flags=int(rng.integers(0, 8)),
                               body=rng.integers(0, 256, nb, dtype=np.uint8).tobytes(), llid=int(rng.integers(0, 4)),
                               flow=int(rng.integers(0, 2)), voice=rng.integers(0, 256, 10, dtype=np.uint8).tobytes(),
                               fhs_bits=synth.fhs_payload(lap, uap, 0x1234, i, rng))[:bt.MAX_SYMBOLS]
        p = p.copy()
        ne = (0, 0, 1, 2, 4, 9)[i % 6]
        if ne and len(p) > 140:
            p[rng.integers(126, len(p), ne)] ^= 1              # payload region: the header stays decodable
        if pos[st] + len(p) + 300 > n_words * 64:
            continue
        sym[st, pos[st]:pos[st] + len(p)] = p
        rows.append((st, pos[st], dict(lap=lap, uap=uap, clk6=clk6, type=t, nbody=nb)))
        # now and then the next packet starts inside this one (a capture cut short by another access code is still
        # decoded to its full window), otherwise a gap of noise
        pos[st] += len(p) + int(rng.integers(1, 120)) if i % 11 else max(200, len(p) // 2)
    return sym, rows


def test_long_payloads_leave_through_the_wave_phase():
    """DM3 / DH3 / DM5 / DH5 payloads beyond 256 bits are not walked by a lane but decoded by a group of 8 .. 64 lanes
    (long_payloads in packet.hip): byte-identical to cutting the packets out and decoding them lane by lane, and equal
    to the oracle -- full lengths, every payload_length mod 8 around the word boundaries, failing FEC 2/3 blocks (the
    reference then writes nothing), wrong clocks (noise through the payload header), unwhitened packets, captures cut
    short by the stream's end and by max_length, and records that hold random bytes on entry."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(83))
    n_streams, n_words = 6, 1 << 15
    sym, rows = _long_capture(rng, 2400, n_streams, n_words)
    # hits whose window runs into the end of the stream, inside long packets put there for it
    for st in range(n_streams):
        for back, t in ((2900, synth.TYPE_DH5), (2000, synth.TYPE_DM5), (1200, synth.TYPE_DH3), (700, synth.TYPE_DM3), (400, synth.TYPE_DH5)):
            lap, uap, clk6 = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 64))
            p = synth.build_packet(lap, uap, clk6, t, lt_addr=1, body=rng.integers(0, 256, LONG_MAXBODY[t], dtype=np.uint8).tobytes())
            off = n_words * 64 - back - 64 * st
            sym[st, off:off + len(p)] = p[:n_words * 64 - off]
            rows.append((st, off, dict(lap=lap, uap=uap, clk6=clk6, type=t, nbody=LONG_MAXBODY[t])))
    assert len(rows) > 1800
    hits = np.zeros(len(rows), bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(len(rows), bt.PKTIN_DTYPE)
    clk = np.array([r[2]["clk6"] for r in rows], dtype=np.uint32)
    clk[::13] ^= 9                                            # wrong clock: the header check decides; some pass by chance
    pin["clkn"] = clk | (rng.integers(0, 1 << 20, len(rows)).astype(np.uint32) << 6)
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    pin["flags"][::29] &= ~np.uint32(1)                       # not whitened
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    dirty = np.frombuffer(rng.integers(0, 256, len(rows) * bt.PKTOUT_DTYPE.itemsize, dtype=np.uint8).tobytes(),
                          dtype=bt.PKTOUT_DTYPE).copy()
    dirty["payload_length"] &= 0x1ff
    dirty["payload_header_length"] &= 3
    for max_length in (bt.MAX_SYMBOLS, 1500):
        direct, len_d = bt.run_decode_hits(words, hits, pin, init_out=dirty, max_length=max_length)
        two_step, len_g = bt.run_decode_hits(words, hits, pin, via_gather=True, init_out=dirty, max_length=max_length)
        assert np.array_equal(len_d, len_g)
        bad = [i for i in range(len(rows)) if direct[i].tobytes() != two_step[i].tobytes()]
        assert not bad, (max_length, len(bad), [(i, rows[i][2]["type"], rows[i][2]["nbody"], int(direct[i]["payload_rv"]),
                                                 int(two_step[i]["payload_rv"]), int(direct[i]["payload_length"])) for i in bad[:8]])
    direct, len_d = bt.run_decode_hits(words, hits, pin)      # records zeroed on entry, as the oracle's packets are
    seen = {}
    for i in range(0, len(rows), 3):
        st, off, meta = rows[i]
        s = np.ascontiguousarray(sym[st, off:off + int(len_d[i])])
        # (the oracle unwhitens whenever asked to decode: only whitened packets are compared with it)
        if not int(pin["flags"][i]) & 1:
            continue
        present, h, r, stt = _both_decode(orc, s, int(pin["clkn"][i]), int(pin["uap"][i]))
        o = direct[i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == (present, h, r), (i, off, meta)
        if h:
            assert int(o["payload_length"]) == stt["payload_length"], (i, meta)
            bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
            assert (bits == stt["payload"]).all(), (i, meta, np.nonzero(bits != stt["payload"])[0][:8])
            seen[(int(o["type"]), r)] = seen.get((int(o["type"]), r), 0) + 1
    for t in LONG_TYPES:
        assert seen.get((t, 10), 0) >= 15, (t, seen)
    assert sum(v for (t, r), v in seen.items() if t in (synth.TYPE_DM3, synth.TYPE_DM5) and r == 0) >= 10, seen
    assert sum(v for (t, r), v in seen.items() if t in LONG_TYPES and r == 2) >= 10, seen


def test_dm5_of_128_bytes_keeps_its_spare_block_bits_to_itself():
    """A DM5 with payload_length 128 (body 124 + header 2 + CRC 2) fills exactly the sixteen packed words of a group of eight
    lanes; its 103rd FEC 2/3 block carries six bits behind the payload.  When that block is hit by two symbol errors
    (mis-corrected or failing), those bits are not zero -- they must not reach word 0 of the packet the next group of the
    wave decodes (round-4 advisor finding).  Every packet of the batch has this length, every other one has the errors; the
    clean neighbours must come out with their CRC intact and every payload bit equal to the oracle's."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(87))
    n_streams, n_words, per_stream = 4, 1 << 14, 96
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    rows = []
    for st in range(n_streams):
        pos = 100 + st
        for i in range(per_stream):
            lap, uap, clk6 = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 64))
            p = synth.build_packet(lap, uap, clk6, synth.TYPE_DM5, lt_addr=1 + i % 7, body=rng.integers(0, 256, 124, dtype=np.uint8).tobytes()).copy()
            # payload symbols start at 122 (sync word 64 + trailer 4 + header 54); block 102 = symbols 122 + 15 * 102 .. + 14, its data
            # bits 4 .. 9 lie behind payload_length
            hurt = (i + st) % 2 == 1
            if hurt:
                b0 = 122 + 15 * 102
                assert len(p) == b0 + 15
                k = rng.choice(np.arange(4, 15), 2, replace=False)
                p[b0 + k] ^= 1
            assert pos + len(p) + 400 < n_words * 64
            sym[st, pos:pos + len(p)] = p
            rows.append((st, pos, dict(uap=uap, clk6=clk6, hurt=hurt)))
            pos += len(p) + int(rng.integers(3, 90))
    hits = np.zeros(len(rows), bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(len(rows), bt.PKTIN_DTYPE)
    pin["clkn"] = [r[2]["clk6"] for r in rows]
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    out, lens = bt.run_decode_hits(words, hits, pin)
    two_step, _ = bt.run_decode_hits(words, hits, pin, via_gather=True)
    clean_ok = spoiled = 0
    for i, (st, off, meta) in enumerate(rows):
        assert out[i].tobytes() == two_step[i].tobytes(), (i, meta)
        s = np.ascontiguousarray(sym[st, off:off + int(lens[i])])
        present, h, r, stt = _both_decode(orc, s, int(pin["clkn"][i]), int(pin["uap"][i]))
        o = out[i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == (present, h, r), (i, meta)
        assert h and int(o["payload_length"]) == stt["payload_length"] == 128, (i, meta)
        bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
        assert (bits == stt["payload"]).all(), (i, meta, np.nonzero(bits != stt["payload"])[0][:8])
        if meta["hurt"]:
            spoiled += r != 10
        else:
            assert r == 10, (i, meta)
            clean_ok += 1
    assert clean_ok >= 150 and spoiled >= 20, (clean_ok, spoiled)


def test_ev4_ev5_payloads_leave_through_the_wave_phase():
    """EV4 / EV5 payloads in HBM are decoded by lane groups too (ev_payloads in packet.hip: the byte count whose CRC register
    is zero comes from a prefix over the lanes, not from a walk): byte-identical to cutting the packets out and decoding
    them lane by lane, and equal to the oracle -- every body length, symbol errors (EV4: blocks that do not decode, in the
    first 45 symbols and behind them), wrong clocks, captures cut short by max_length and by the next packet, records that
    hold random bytes on entry, and DM / DH packets in the same waves."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(1213))
    n_streams, n_words = 6, 4096
    sym, rows = _long_capture(rng, 2400, n_streams, n_words, others=(synth.TYPE_DM3, synth.TYPE_DH5, synth.TYPE_DH1, synth.TYPE_DM1),
                              longs=(synth.TYPE_EV4, synth.TYPE_EV5, synth.TYPE_EV5, synth.TYPE_EV4))
    assert len(rows) > 1400
    # errors in the first three blocks of some EV4 packets (the reference answers 0 there, 1 behind them)
    for j, (st, off, meta) in enumerate(rows):
        if meta["type"] == synth.TYPE_EV4 and j % 9 == 0:
            sym[st, off + 122 + int(rng.integers(0, 45))] ^= 1
            sym[st, off + 122 + int(rng.integers(0, 45))] ^= 1
    hits = np.zeros(len(rows), bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(len(rows), bt.PKTIN_DTYPE)
    clk = np.array([r[2]["clk6"] for r in rows], dtype=np.uint32)
    clk[::13] ^= 9
    pin["clkn"] = clk | (rng.integers(0, 1 << 20, len(rows)).astype(np.uint32) << 6)
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    pin["flags"][::29] &= ~np.uint32(1)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    dirty = np.frombuffer(rng.integers(0, 256, len(rows) * bt.PKTOUT_DTYPE.itemsize, dtype=np.uint8).tobytes(),
                          dtype=bt.PKTOUT_DTYPE).copy()
    dirty["payload_length"] &= 0x1ff
    dirty["payload_header_length"] &= 3
    for max_length in (bt.MAX_SYMBOLS, 900, 400, 140):
        direct, len_d = bt.run_decode_hits(words, hits, pin, init_out=dirty, max_length=max_length)
        two_step, len_g = bt.run_decode_hits(words, hits, pin, via_gather=True, init_out=dirty, max_length=max_length)
        assert np.array_equal(len_d, len_g)
        bad = [i for i in range(len(rows)) if direct[i].tobytes() != two_step[i].tobytes()]
        assert not bad, (max_length, len(bad), [(i, rows[i][2]["type"], rows[i][2]["nbody"], int(direct[i]["payload_rv"]),
                                                 int(two_step[i]["payload_rv"]), int(direct[i]["payload_length"]),
                                                 int(two_step[i]["payload_length"])) for i in bad[:8]])
    direct, len_d = bt.run_decode_hits(words, hits, pin)
    seen = {}
    for i in range(0, len(rows), 2):
        st, off, meta = rows[i]
        if not int(pin["flags"][i]) & 1:
            continue
        s = np.ascontiguousarray(sym[st, off:off + int(len_d[i])])
        present, h, r, stt = _both_decode(orc, s, int(pin["clkn"][i]), int(pin["uap"][i]))
        o = direct[i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == (present, h, r), (i, off, meta)
        if h:
            assert int(o["payload_length"]) == stt["payload_length"], (i, meta)
            bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
            assert (bits == stt["payload"]).all(), (i, meta, np.nonzero(bits != stt["payload"])[0][:8])
            seen[(int(o["type"]), r)] = seen.get((int(o["type"]), r), 0) + 1
    assert seen.get((synth.TYPE_EV4, 10), 0) >= 40, sorted(seen.items())
    assert seen.get((synth.TYPE_EV4, 0), 0) >= 5 and seen.get((synth.TYPE_EV4, 1), 0) >= 5, sorted(seen.items())
    # (EV5: the reference's loop repeats the first payload byte -- SURVEY Q7 --, its CRC matches by chance only)
    assert seen.get((synth.TYPE_EV5, 1), 0) + seen.get((synth.TYPE_EV5, 2), 0) >= 40, sorted(seen.items())


def test_ev5_lengths_found_by_the_prefix_over_the_lanes():
    """The reference's EV5 loop repeats the first payload byte under the whitening of each place, so where its
    CRC register reaches zero -- payload_length, verdict 10, the bytes written -- is decided by (first byte, CLK1-6, UAP)
    alone, once in ~370 packets.  Thousands of EV5 headers in front of noise: every record equal to the lane-by-lane decode of
    the cut-out packet, and the ones that end early equal to the oracle."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(77))
    n_streams, per_stream, gap = 8, 768, 1700
    n_words = (per_stream * gap + 4096) // 64
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    rows = []
    for st in range(n_streams):
        for k in range(per_stream):
            lap, uap, clk6 = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 64))
            p = synth.build_packet(lap, uap, clk6, synth.TYPE_EV5, lt_addr=1 + k % 7, body=b"")[:122]
            off = 64 + k * gap + int(rng.integers(0, 64))
            sym[st, off:off + len(p)] = p
            rows.append((st, off, uap, clk6))
    n = len(rows)
    hits = np.zeros(n, bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(n, bt.PKTIN_DTYPE)
    pin["clkn"] = [r[3] for r in rows]
    pin["uap"] = [r[2] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    dirty = np.frombuffer(rng.integers(0, 256, n * bt.PKTOUT_DTYPE.itemsize, dtype=np.uint8).tobytes(), dtype=bt.PKTOUT_DTYPE).copy()
    dirty["payload_length"] &= 0x1ff
    dirty["payload_header_length"] &= 3
    for max_length in (bt.MAX_SYMBOLS, 122 + 8 * 90):
        direct, len_d = bt.run_decode_hits(words, hits, pin, init_out=dirty, max_length=max_length)
        two_step, len_g = bt.run_decode_hits(words, hits, pin, via_gather=True, init_out=dirty, max_length=max_length)
        assert np.array_equal(len_d, len_g)
        bad = [i for i in range(n) if direct[i].tobytes() != two_step[i].tobytes()]
        assert not bad, (max_length, len(bad), [(i, int(direct[i]["payload_rv"]), int(two_step[i]["payload_rv"]),
                                                 int(direct[i]["payload_length"]), int(two_step[i]["payload_length"])) for i in bad[:8]])
    direct, len_d = bt.run_decode_hits(words, hits, pin)
    assert (direct["type"] == synth.TYPE_EV5).sum() > n - 10
    early = np.nonzero((direct["payload_rv"] == 10) & (direct["type"] == synth.TYPE_EV5))[0]
    assert len(early) >= 6, len(early)
    for i in list(early) + list(range(0, n, 97)):
        st, off, uap, clk6 = rows[i]
        s = np.ascontiguousarray(sym[st, off:off + int(len_d[i])])
        present, h, r, stt = _both_decode(orc, s, clk6, uap)
        o = direct[i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == (present, h, r), (i, rows[i])
        assert int(o["payload_length"]) == stt["payload_length"], (i, rows[i])
        bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
        assert (bits == stt["payload"]).all(), (i, rows[i])


def test_decode_with_the_count_in_hbm():
    """btbbx_decode_hits_counted_device (the list's length is a word in HBM, the launch is sized for the capacity):
    the first min(count, capacity) records equal btbbx_decode_hits_device's, records behind the count keep every byte
    they held -- for a count inside a workgroup, on a workgroup boundary, equal to and above the capacity."""
    rng = np.random.default_rng(_libs.seed(89))
    n_streams, n_words = 3, 1 << 14
    sym, rows = _long_capture(rng, 700, n_streams, n_words)
    n = len(rows)
    assert n > 520
    hits = np.zeros(n, bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(n, bt.PKTIN_DTYPE)
    pin["clkn"] = [r[2]["clk6"] for r in rows]
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    dirty = np.frombuffer(rng.integers(0, 256, n * bt.PKTOUT_DTYPE.itemsize, dtype=np.uint8).tobytes(), dtype=bt.PKTOUT_DTYPE).copy()
    dirty["payload_length"] &= 0x1ff
    dirty["payload_header_length"] &= 3
    full, len_f = bt.run_decode_hits(words, hits, pin, init_out=dirty)
    for count in (0, 1, 77, 256, 300, 512, n - 1, n, n + 5, 1 << 31):
        got, len_c = bt.run_decode_hits(words, hits, pin, init_out=dirty, count=count)
        k = min(count, n)
        assert got[:k].tobytes() == full[:k].tobytes(), count
        assert got[k:].tobytes() == dirty[k:].tobytes(), count
        assert np.array_equal(len_c[:k], len_f[:k]) and not len_c[k:].any(), count


def test_piconet_decode_entry_equals_the_per_packet_form():
    """btbbx_decode_hits_piconet_device (one entry state for all packets, clock = entry.clkn + offset / clk_div worked out in the
    kernel) against btbbx_decode_hits_counted_device with the same clocks written into a btbbx_pkt_in per packet: byte-identical
    records -- with a divisor above the stream length (every packet at the entry's clock: the packets were built for it and decode),
    with 625 and 4096 (clocks from the offsets), with and without the count in HBM."""
    lib = bt.lib()
    rng = np.random.default_rng(_libs.seed(97))
    n_streams, n_words = 4, 1 << 14
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    lap, uap, clk6 = 0x5A17C3, 0x6B, 0x19
    rows, pos = [], [100 + int(rng.integers(0, 64)) for _ in range(n_streams)]
    types = (synth.TYPE_DM1, synth.TYPE_DH1, synth.TYPE_DM3, synth.TYPE_DH3, synth.TYPE_DM5, synth.TYPE_DH5, synth.TYPE_FHS, synth.TYPE_HV3)
    for i in range(900):
        st, t = i % n_streams, types[(i // n_streams) % len(types)]
        nb = 30 if t == synth.TYPE_HV3 else int(rng.integers(0, LONG_MAXBODY.get(t, 0) + 1))
        p = synth.build_packet(lap, uap, clk6, t, lt_addr=1 + i % 7, flags=i % 8, body=rng.integers(0, 256, nb, dtype=np.uint8).tobytes(),
                               fhs_bits=synth.fhs_payload(lap, uap, 0x1234, i, rng))[:bt.MAX_SYMBOLS]
        if pos[st] + len(p) + 300 > n_words * 64:
            continue
        sym[st, pos[st]:pos[st] + len(p)] = p
        rows.append((st, pos[st]))
        pos[st] += len(p) + int(rng.integers(1, 150))
    n = len(rows)
    assert n > 400
    hits = np.zeros(n, bt.HIT_DTYPE)
    hits["stream"], hits["offset"] = [r[0] for r in rows], [r[1] for r in rows]
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    flags = (1 << 0) | (1 << 2) | (1 << 4)
    entry = np.zeros(1, bt.PKTIN_DTYPE)
    d_w = bt.DeviceBuffer(words.nbytes).upload(words)
    d_h = bt.DeviceBuffer(hits.nbytes).upload(hits)
    good = 0
    for clk_div, base, count in ((1 << 30, clk6, None), (625, 5, n), (4096, 0x123456, n - 7)):
        entry["clkn"], entry["flags"], entry["uap"] = base, flags, uap
        pin = np.zeros(n, bt.PKTIN_DTYPE)
        pin["clkn"] = (base + hits["offset"] // clk_div).astype(np.uint32)
        pin["flags"], pin["uap"] = flags, uap
        want, len_w = bt.run_decode_hits(words, hits, pin, count=count)
        d_out = bt.DeviceBuffer(n * bt.PKTOUT_DTYPE.itemsize).zero()
        d_len = bt.DeviceBuffer(n * 4).zero()
        d_cnt = bt.DeviceBuffer(8).upload(np.array([count or 0, 0], dtype=np.uint32))
        bt.check(lib.btbbx_decode_hits_piconet_device(d_w.ptr, n_words, n_words, d_h.ptr, d_cnt.ptr if count is not None else None, n,
                                                      entry.ctypes.data_as(C.c_void_p), clk_div, bt.MAX_SYMBOLS, d_out.ptr, d_len.ptr, None))
        bt.check(lib.btbbx_sync(None))
        got = d_out.download(bt.PKTOUT_DTYPE, n)
        assert got.tobytes() == want.tobytes(), (clk_div, [i for i in range(n) if got[i].tobytes() != want[i].tobytes()][:5])
        assert np.array_equal(d_len.download(np.uint32, n), len_w)
        good = max(good, int((got["payload_rv"] == 10).sum() + (got["payload_rv"] == 1000).sum()))
        for b in (d_out, d_len, d_cnt):
            b.free()
    assert good > 300                                             # the packets built for the entry's clock decode with it
    assert lib.btbbx_decode_hits_piconet_device(d_w.ptr, n_words, n_words, d_h.ptr, None, n, entry.ctypes.data_as(C.c_void_p), 0,
                                                bt.MAX_SYMBOLS, d_w.ptr, None, None) < 0          # clk_div = 0 is refused
    # (round 5) a buffer that starts `phase` symbols into a slot: clock = entry.clkn + (offset + phase) / clk_div; the entry's
    # length field, which the kernel uses to carry the phase, is ignored whatever the caller left in it
    for clk_div, phase, base in ((625, 0, 77), (625, 1, 77), (625, 624, 77), (4096, 3000, 0x3FFFFF0)):
        entry["clkn"], entry["flags"], entry["uap"], entry["length"] = base, flags, uap, 12345
        pin = np.zeros(n, bt.PKTIN_DTYPE)
        pin["clkn"] = ((base + (hits["offset"] + np.uint64(phase)) // np.uint64(clk_div)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        pin["flags"], pin["uap"] = flags, uap
        want, _ = bt.run_decode_hits(words, hits, pin)
        d_out = bt.DeviceBuffer(n * bt.PKTOUT_DTYPE.itemsize).zero()
        bt.check(lib.btbbx_decode_hits_piconet_phase_device(d_w.ptr, n_words, n_words, d_h.ptr, None, n, entry.ctypes.data_as(C.c_void_p),
                                                            clk_div, phase, bt.MAX_SYMBOLS, d_out.ptr, None, None), "piconet_phase")
        bt.check(lib.btbbx_sync(None))
        got = d_out.download(bt.PKTOUT_DTYPE, n)
        assert got.tobytes() == want.tobytes(), (clk_div, phase, [i for i in range(n) if got[i].tobytes() != want[i].tobytes()][:5])
        d_out.free()
    assert lib.btbbx_decode_hits_piconet_phase_device(d_w.ptr, n_words, n_words, d_h.ptr, None, n, entry.ctypes.data_as(C.c_void_p), 625,
                                                      625, bt.MAX_SYMBOLS, d_w.ptr, None, None) < 0  # phase >= clk_div is refused
    d_w.free()
    d_h.free()


def test_decode_from_the_streams_equals_gather_then_decode():
    """btbbx_decode_hits_device reads the packets where they lie: same btbbx_pkt_out, byte for byte, as cutting
    them out first -- for every bit alignment, for captures cut short by the end of the stream (the decoders
    read zeros behind the captured length in both) and for hits in the very last words; and the oracle on the
    same symbols agrees."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(53))
    n_streams, n_words = 3, 4096
    sym = rng.integers(0, 2, (n_streams, n_words * 64), dtype=np.uint8)
    pk = _pkt.random_packets(rng, 150, max_sym_errors=2)
    hits = np.zeros(0, bt.HIT_DTYPE)
    rows = []
    pos = [64 + int(rng.integers(0, 64)) for _ in range(n_streams)]
    for i, (s, meta) in enumerate(pk):
        st = i % n_streams
        s = s[:bt.MAX_SYMBOLS]
        if pos[st] + len(s) + 200 > n_words * 64:
            continue
        sym[st, pos[st]:pos[st] + len(s)] = s
        rows.append((st, pos[st], meta))
        pos[st] += len(s) + int(rng.integers(1, 190))
    # hits whose capture window runs into the end of the stream, down to a window of a few symbols
    for st in range(n_streams):
        for back in (3124, 3000, 1500, 400, 130, 121, 70, 1):
            rows.append((st, n_words * 64 - back, dict(lap=0, uap=int(rng.integers(0, 256)), clk6=int(rng.integers(0, 64)), type=-1)))
    hits = np.zeros(len(rows), bt.HIT_DTYPE)
    hits["stream"] = [r[0] for r in rows]
    hits["offset"] = [r[1] for r in rows]
    pin = np.zeros(len(rows), bt.PKTIN_DTYPE)
    pin["clkn"] = [r[2]["clk6"] for r in rows]
    pin["uap"] = [r[2]["uap"] for r in rows]
    pin["flags"] = (1 << 0) | (1 << 2) | (1 << 4)
    words = np.stack([synth.pack_bits(sym[st]) for st in range(n_streams)])
    direct, len_d = bt.run_decode_hits(words, hits, pin)
    two_step, len_g = bt.run_decode_hits(words, hits, pin, via_gather=True)
    assert np.array_equal(len_d, len_g)
    assert len_d.min() == 1 and (len_d == bt.MAX_SYMBOLS).sum() > 50
    assert direct.tobytes() == two_step.tobytes()
    good = 0
    for i, (st, off, meta) in enumerate(rows):
        s = np.ascontiguousarray(sym[st, off:off + int(len_d[i])])
        present, h, r, stt = _oracle_decode(orc, s, int(pin["clkn"][i]), int(pin["uap"][i]))
        o = direct[i]
        assert (int(o["header_present"]), int(o["header_rv"]), int(o["payload_rv"])) == (present, h, r), (i, off, meta)
        if h:
            assert int(o["payload_length"]) == stt["payload_length"], (i, meta)
            bits = synth.unpack_bits(np.ascontiguousarray(o["payload"]), 2744)
            assert (bits == stt["payload"]).all(), (i, meta)
            good += r in (10, 1000)
    assert good > 20


class DropIn:
    """The same packet in the product (C ABI) and in the oracle."""

    def __init__(self, lib, orc, lap):
        self.lib, self.orc = lib, orc
        self.p = C.c_void_p(lib.btbb_packet_new())
        self.o = orc.orc_packet_new()
        orc.orc_packet_init_found(self.o, lap, 0)
        lib.btbb_packet_set_flag(self.p, 0, 1)

    def set_data(self, sym, channel, clkn):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        self.lib.btbb_packet_set_data(self.p, _libs.ptr(sym), len(sym), channel, clkn)
        self.orc.orc_packet_set_data(self.o, _libs.ptr(sym), len(sym), channel, clkn)

    def check(self, ctx=""):
        lib, p, st = self.lib, self.p, _pkt.orc_state(self.o)
        for f in range(15):
            assert lib.btbb_packet_get_flag(p, f) == ((st["flags"] >> f) & 1), (ctx, "flag", f)
        assert lib.btbb_packet_get_uap(p) == st["UAP"], ctx
        assert lib.btbb_packet_get_type(p) == st["packet_type"], ctx
        assert lib.btbb_packet_get_lt_addr(p) == st["packet_lt_addr"], ctx
        assert lib.btbb_packet_get_header_flags(p) == st["packet_flags"], ctx
        assert lib.btbb_packet_get_hec(p) == st["packet_hec"], ctx
        assert lib.btbb_packet_get_header_packed(p) == int(sum(int(b) << k for k, b in enumerate(st["packet_header"]))), ctx
        assert lib.btbb_packet_get_payload_length(p) == st["payload_length"], ctx
        assert lib.btbb_packet_get_clkn(p) == st["clkn"], ctx
        pay = np.frombuffer((C.c_uint8 * 2744).from_address(lib.btbb_get_payload(p)), dtype=np.uint8)
        assert (pay == st["payload"]).all(), (ctx, "payload", np.nonzero(pay != st["payload"])[0][:8])

    def close(self):
        self.lib.btbb_packet_unref(self.p)
        self.orc.orc_packet_free(self.o)


def test_drop_in_decode(capfd):
    lib, orc = bt.lib(), _libs.oracle()
    rng = np.random.default_rng(_libs.seed(33))
    ok = 0
    for sym, meta in _pkt.random_packets(rng, 60, max_sym_errors=1):
        d = DropIn(lib, orc, meta["lap"])
        clkn = (int(rng.integers(0, 1 << 20)) << 7) | (meta["clk6"] << 1)
        d.set_data(sym, 5, clkn)
        lib.btbb_packet_set_uap(d.p, meta["uap"])
        d.o.contents.UAP = meta["uap"]
        orc.orc_packet_set_flag(d.o, 2, 1)
        assert lib.btbb_header_present(d.p) == orc.orc_header_present(d.o)
        assert lib.btbb_decode_header(d.p) == orc.orc_decode_header(d.o) == 0
        lib.btbb_packet_set_flag(d.p, 4, 1)
        orc.orc_packet_set_flag(d.o, 4, 1)
        h1, h2 = lib.btbb_decode_header(d.p), orc.orc_decode_header(d.o)
        assert h1 == h2
        d.check(("hdr", meta))
        if h1:
            r1, r2 = lib.btbb_decode_payload(d.p), orc.orc_decode_payload(d.o)
            assert r1 == r2
            ok += r1 in (10, 1000)
            d.check(("payload", meta))
            b1, b2 = np.zeros(400, np.uint8), np.zeros(400, np.uint8)
            assert lib.btbb_get_payload_packed(d.p, _libs.ptr(b1)) == orc.orc_payload_packed(d.o, _libs.ptr(b2))
            assert (b1 == b2).all()
        assert lib.btbb_decode(d.p) == orc.orc_decode(d.o)
        d.check(("decode", meta))
        d.close()
    capfd.readouterr()
    assert ok > 8


def test_drop_in_try_clock_crc_check_and_fhs_fields(capfd):
    """The internal entry points the reference library also exports: try_clock / crc_check one
    clock at a time on the same packet object (state carried exactly like the oracle), FHS
    field extractors and tun_format after a decode."""
    lib, orc = bt.lib(), _libs.oracle()
    rng = np.random.default_rng(_libs.seed(35))
    for sym, meta in _pkt.random_packets(rng, 24, max_sym_errors=1):
        d = DropIn(lib, orc, meta["lap"])
        d.set_data(sym, 7, meta["clk6"] << 1)
        for clock in (meta["clk6"], (meta["clk6"] + 9) % 64, 63):
            assert lib.try_clock(clock, d.p) == orc.orc_try_clock(clock, d.o)
            assert lib.crc_check(clock, d.p) == orc.orc_crc_check(clock, d.o)
            d.check((meta, clock))
        d.close()
    lap, uap = 0x654321, 0x5A
    fb = synth.fhs_payload(lap, uap, 0xBEEF, 0x2345678, rng)
    sym = synth.build_packet(lap, uap, 11, synth.TYPE_FHS, fhs_bits=fb)
    d = DropIn(lib, orc, lap)
    d.set_data(sym, 3, 11 << 1)
    lib.btbb_packet_set_uap(d.p, uap)
    lib.btbb_packet_set_flag(d.p, 4, 1)
    assert lib.btbb_decode(d.p) == 1000
    assert lib.lap_from_fhs(d.p) == lap and lib.uap_from_fhs(d.p) == uap
    assert lib.nap_from_fhs(d.p) == 0xBEEF and lib.clock_from_fhs(d.p) == 0x2345678
    tun = lib.tun_format(d.p)
    raw = bytes((C.c_uint8 * (9 + 20)).from_address(tun))
    assert raw[4] == 3 and raw[8] == lib.btbb_packet_get_hec(d.p)
    pk = np.zeros(64, np.uint8)
    lib.btbb_get_payload_packed(d.p, _libs.ptr(pk))
    assert raw[9:29] == bytes(pk[:20])
    C.CDLL(None).free(C.c_void_p(tun))
    d.close()
    capfd.readouterr()


def test_drop_in_uap_from_header(capfd):
    """Piconet UAP / CLK1-6 discovery over packet sequences: return values, piconet state and
    the packet object after each call equal the oracle's."""
    lib, orc = bt.lib(), _libs.oracle()
    rng = np.random.default_rng(_libs.seed(34))
    found = 0
    for seq in range(12):
        lap, uap = int(rng.integers(0, 1 << 24)), int(rng.integers(1, 256))
        pn = C.c_void_p(lib.btbb_piconet_new())
        on = orc.orc_piconet_new()
        lib.btbb_init_piconet(pn, lap)
        orc.orc_init_piconet(on, lap)
        clk = int(rng.integers(0, 1 << 26))
        for k in range(10):
            clk += int(rng.integers(1, 40)) * 2
            clk6 = (clk >> 1) & 0x3F
            t = [0, 1, 9, 6][k % 4] if k < int(rng.integers(2, 7)) else [3, 4, 10, 2][k % 4]
            body = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
            sym = synth.build_packet(lap, uap, clk6, t, lt_addr=1, body=body, fhs_bits=synth.fhs_payload(lap, uap, 1, 2, rng))
            sym = np.concatenate([sym, rng.integers(0, 2, 50, dtype=np.uint8)])
            if rng.random() < 0.3:
                sym[int(rng.integers(68, len(sym)))] ^= 1
            d = DropIn(lib, orc, lap)
            d.set_data(sym, int(rng.integers(0, 79)), (clk + 34) & 0xFFFFFFF)
            if seq % 2:
                a, b = lib.btbb_uap_from_header(d.p, pn), orc.orc_uap_from_header(d.o, on)
            else:
                a, b = lib.btbb_process_packet(d.p, pn), orc.orc_process_packet(d.o, on)
            assert a == b, (seq, k)
            d.check((seq, k))
            c = on.contents
            for f in range(15):
                assert lib.btbb_piconet_get_flag(pn, f) == ((c.flags >> f) & 1), (seq, k, f)
            assert lib.btbb_piconet_get_uap(pn) == c.UAP and lib.btbb_piconet_get_clk_offset(pn) == c.clk_offset
            d.close()
            if orc.orc_piconet_get_flag(on, 2) and orc.orc_piconet_get_flag(on, 4):
                found += c.UAP == uap
                if seq % 2 == 0:
                    break
        lib.btbb_piconet_unref(pn)
        orc.orc_piconet_free(on)
    capfd.readouterr()
    assert found >= 7


def test_uap_table():
    """btbbx_uap_table_device (one LFSR run per packet + a constant per clock, by linearity) equals
    try_clock for all 64 clocks: return value and the packet type it leaves."""
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(36))
    pk = _pkt.random_packets(rng, 700)
    syms = [np.ascontiguousarray(s[:bt.MAX_SYMBOLS]) for s, _ in pk]
    for k in range(0, len(syms), 9):                    # some headers beyond FEC 1/3's reach
        if len(syms[k]) > 130:
            syms[k][68 + rng.integers(0, 54, int(rng.integers(3, 9)))] ^= 1
    words, lengths = bt.packets_to_words(syms)
    pin = np.zeros(len(syms), bt.PKTIN_DTYPE)
    pin["length"] = lengths
    pin["flags"] = 1
    pin["flags"][::13] = 0                              # a few unwhitened
    got = bt.run_uap_table(words, pin)
    assert got.shape == (len(syms), 64)
    zeros = 0
    for i, s in enumerate(syms):
        p = orc.orc_packet_new()
        orc.orc_packet_init_found(p, 0, 0)
        orc.orc_packet_set_flag(p, 0, int(pin["flags"][i]) & 1)
        orc.orc_packet_set_data(p, _libs.ptr(s), len(s), 0, 0)
        for clock in range(64):
            p.contents.packet_type = 0
            u = orc.orc_try_clock(clock, p)
            want = u | (p.contents.packet_type << 8)
            assert int(got[i, clock]) == want, (i, clock, hex(int(got[i, clock])), hex(want))
        zeros += int((got[i] == 0).all())
        orc.orc_packet_free(p)
    assert 20 < zeros < 200                             # FEC failures present, but not the rule
    # no pkt_in means "all whitened"
    allw = pin[:100].copy()
    allw["flags"] = 1
    assert np.array_equal(bt.run_uap_table(words[:100]), bt.run_uap_table(words[:100], allw))
