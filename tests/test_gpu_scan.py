"""GPU parity: the HIP access-code scan (through the C ABI) vs the oracle on the same seeded
streams -- bit-exact hit lists (offset, LAP, ac_errors).  Run with `-m gpu` on an MI355X."""
import ctypes as C

import numpy as np
import pytest

import _libs
import libbtbb_amd as bt
from libbtbb_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def ready():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    bt.lib().btbbx_shutdown()
    bt.init(2)
    orc = _libs.oracle()
    orc.orc_reset_syndrome_map()
    orc.orc_init(2)
    yield
    bt.lib().btbbx_shutdown()


def as_tuples(hits):
    return [(int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in hits]


def stream(seed, nwords, **kw):
    words, inj = synth.make_stream(seed, nwords, **kw)
    return words, np.ascontiguousarray(synth.unpack_bits(words)), inj


@pytest.mark.parametrize("max_err", [0, 1, 2, 3])
def test_lap_any_parity(max_err):
    words, sym, inj = stream(101, 1 << 14, stride=1024)          # 1 Mi symbols, 1024 injections
    n = len(sym) - 63
    got = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, max_err))
    want = _libs.orc_find_all(sym, n, _libs.LAP_ANY, max_err)
    assert got == want
    assert len(got) > 200


@pytest.mark.parametrize("max_err", [0, 1, 2, 3, 5, 9])
def test_known_lap_parity(max_err):
    lap = 0x9E8B33
    words, sym, inj = stream(102, 1 << 14, stride=1024, lap=lap)
    n = len(sym) - 63
    got = as_tuples(bt.scan_words(words, n, lap, max_err))
    want = _libs.orc_find_all(sym, n, lap, max_err)
    assert got == want and len(got) > 200


def test_negative_and_edge_arguments():
    words, sym, _ = stream(103, 256, stride=512)
    assert len(bt.scan_words(words, 1000, bt.LAP_ANY, -1)) == 0
    assert len(bt.scan_words(words, 1000, 0x123456, -1)) == 0
    assert len(bt.scan_words(words, 0, bt.LAP_ANY, 2)) == 0
    with pytest.raises(bt.BtbbError):
        bt.scan_words(words, len(words) * 64, bt.LAP_ANY, 2)       # would read past the end
    # ragged search lengths around word / tile boundaries
    for n in (1, 63, 64, 65, 1023, 4096 + 17, len(sym) - 63):
        for lap in (bt.LAP_ANY, 0x9E8B33):
            assert as_tuples(bt.scan_words(words, n, lap, 2)) == _libs.orc_find_all(sym, n, lap, 2), (n, lap)


def test_quirks_barker_and_bit57():
    """barker errors corrected but not counted; bit 57 both ways."""
    lap = 0x654321
    sw = synth.syncword(lap)
    for flips in [(60, 3, 30), (57,), (3, 30, 44), (58, 59), (63,), (57, 3), (0, 1), (56, 55, 54), (61, 2), ()]:
        w = sw
        for b in flips:
            w ^= 1 << b
        for shift in (0, 1, 31, 32, 33, 63):
            sym = np.concatenate([np.zeros(128 + shift, np.uint8), synth.bits_lsb(w, 64), np.zeros(200, np.uint8)])
            words = synth.pack_bits(sym)
            n = len(sym) - 63
            for mode_lap in (bt.LAP_ANY, lap):
                for me in (0, 1, 2, 3):
                    got = as_tuples(bt.scan_words(words, n, mode_lap, me))
                    assert got == _libs.orc_find_all(sym, n, mode_lap, me), (flips, shift, mode_lap, me)


def test_adversarial_all_candidates():
    """A stream made only of sync words back to back: every 64th offset matches and the
    candidate rings overflow into the in-place verification path."""
    rng = np.random.default_rng(5)
    laps = rng.integers(0, 1 << 24, 4096)
    words = np.array([synth.syncword(int(l)) for l in laps], dtype=np.uint64)
    words = np.concatenate([words, np.zeros(1, np.uint64)])
    sym = np.ascontiguousarray(synth.unpack_bits(words))
    n = len(sym) - 63
    got = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, 2))
    assert got == _libs.orc_find_all(sym, n, _libs.LAP_ANY, 2)
    assert len(got) >= 4096


def test_adversarial_two_hits_per_word():
    """Two access codes that START in the same 64-bit word, in every second word: a sync word at bit 0 and one at
    bit 57 whose seven lowest (parity) bits equal the first one's seven highest (barker + LAP MSB) bits.  A lane
    then has four candidates per trip of the scan loop and two private slots: the path that hands a survivor
    back and compacts in the middle of a trip (sliding-check kernel) / checks in place (tables for >= 4 errors)."""
    rng = np.random.default_rng(_libs.seed(77))
    pairs = []
    while len(pairs) < 24:
        l1 = int(rng.integers(0, 1 << 24))
        top7 = synth.syncword(l1) >> 57
        for _ in range(4000):
            l2 = int(rng.integers(0, 1 << 24))
            if synth.syncword(l2) & 0x7F == top7:
                pairs.append((l1, l2))
                break
    words = []
    for k in range(3000):
        s1, s2 = (synth.syncword(l) for l in pairs[k % len(pairs)])
        both = s1 | (s2 << 57)                     # 121 bits: s2's low seven bits coincide with s1's top seven
        words += [both & ((1 << 64) - 1), both >> 64]
    words = np.array(words + [0], dtype=np.uint64)
    sym = np.ascontiguousarray(synth.unpack_bits(words))
    n = len(sym) - 63
    for max_err in (0, 2):
        got = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, max_err))
        assert got == _libs.orc_find_all(sym, n, _libs.LAP_ANY, max_err)
        assert len(got) >= 6000


def test_symbols_entry_and_pack_roundtrip():
    words, sym, _ = stream(104, 1 << 10, stride=512)
    n = len(sym) - 63
    a = as_tuples(bt.scan_symbols(sym, n, bt.LAP_ANY, 2))
    assert a == _libs.orc_find_all(sym, n, _libs.LAP_ANY, 2) and len(a) > 50
    # odd lengths through pack/unpack
    lib = bt.lib()
    big = np.ascontiguousarray(synth.unpack_bits(synth.noise_words(9, 0, 2048)))
    for ns in (1, 15, 16, 17, 63, 64, 65, 1000, 65536 + 3):
        s = np.ascontiguousarray(big[:ns])
        d_s = bt.DeviceBuffer(ns + 64).upload(s)
        d_w = bt.DeviceBuffer(((ns + 63) // 64) * 8 + 8)
        d_o = bt.DeviceBuffer(ns + 64).zero()
        bt.check(lib.btbbx_pack_device(d_s.ptr, ns, d_w.ptr, None))
        bt.check(lib.btbbx_unpack_device(d_w.ptr, ns, d_o.ptr, None))
        bt.check(lib.btbbx_sync(None))
        w = d_w.download(np.uint64, (ns + 63) // 64)
        assert (w == synth.pack_bits(s)).all()
        assert (d_o.download(np.uint8, ns) == s).all()


def test_device_generator_matches_host():
    lib = bt.lib()
    for first, nw, stride, lap, cyc in [(0, 4096, 1024, -1, 4), (777, 5000, 512, 0x9E8B33, 3), (1 << 20, 3000, 4096, -1, 4)]:
        d = bt.DeviceBuffer(nw * 8)
        bt.check(lib.btbbx_synth_device(d.ptr, first, nw, 0xC0FFEE, stride, lap, cyc, None))
        bt.check(lib.btbbx_sync(None))
        got = d.download(np.uint64, nw)
        want, _ = synth.make_stream(0xC0FFEE, nw, stride=stride, first_word=first,
                                    lap=None if lap < 0 else lap, err_cycle=cyc)
        assert (got == want).all()


def test_multi_stream_launch():
    """79 channels in one launch: hits carry their stream index."""
    lib = bt.lib()
    nstreams, nw, pitch = 79, 600, 640
    host = np.zeros(nstreams * pitch, np.uint64)
    want = []
    for s in range(nstreams):
        w, _ = synth.make_stream(1000 + s, nw, stride=512)
        host[s * pitch: s * pitch + nw] = w
        sym = np.ascontiguousarray(synth.unpack_bits(w))
        want += [(s,) + t for t in _libs.orc_find_all(sym, nw * 64 - 63, _libs.LAP_ANY, 2)]
    d_w = bt.DeviceBuffer(host.nbytes).upload(host)
    cap = 1 << 16
    d_h = bt.DeviceBuffer(cap * 16)
    d_c = bt.DeviceBuffer(4).zero()
    bt.check(lib.btbbx_scan_device(d_w.ptr, nw, pitch, nstreams, nw * 64 - 63, bt.LAP_ANY, 2, d_h.ptr, cap, d_c.ptr, None))
    bt.check(lib.btbbx_sync(None))
    cnt = int(d_c.download(np.uint32, 1)[0])
    hits = d_h.download(bt.HIT_DTYPE, cnt)
    lib.btbbx_sort_hits(hits.ctypes.data_as(C.c_void_p), cnt)
    got = [(int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in hits]
    assert got == want and cnt > 2000


def test_init_semantics_first_nonzero_wins():
    """tables come from the first non-zero btbb_init; later values only filter."""
    lib = bt.lib()
    orc = _libs.oracle()
    words, sym, _ = stream(105, 1 << 12, stride=512)
    n = len(sym) - 63
    try:
        for first_init in (0, 1, 3):
            lib.btbbx_shutdown()
            orc.orc_reset_syndrome_map()
            assert lib.btbb_init(first_init) == 0 and orc.orc_init(first_init) == 0
            assert lib.btbb_init(2) == 0 and orc.orc_init(2) == 0
            for me in (0, 1, 2, 3):
                assert as_tuples(bt.scan_words(words, n, bt.LAP_ANY, me)) == _libs.orc_find_all(sym, n, _libs.LAP_ANY, me), (first_init, me)
    finally:
        lib.btbbx_shutdown()
        orc.orc_reset_syndrome_map()
        bt.init(2)
        orc.orc_init(2)


@pytest.mark.parametrize("n_init", [3, 4, 5])
def test_large_error_tables(n_init):
    """btbb_init(3) / btbb_init(4): 32 567 / 457 k error patterns -- the two-level form of scan_slide_kernel (3 % / 32 % of the
    survivors are members of its 2^20-bit set in LDS and look a second check stream up in a set in L2; more in
    test_three_and_four_error_tables_two_level_kernel), on a stream long enough for every wave to drain its ring.
    btbb_init(5): 5.0 M patterns; every survivor probes a bitmap in L2 (scan_lap_any_kernel) -- slow but bit-exact."""
    lib = bt.lib()
    orc = _libs.oracle()
    words, inj = synth.make_stream(108, (1 << 15) + 77 if n_init <= 4 else 1 << 11, stride=512, err_cycle=7)     # 0..5 (+6) bit errors
    sym = np.ascontiguousarray(synth.unpack_bits(words))
    n = len(sym) - 63
    try:
        lib.btbbx_shutdown()
        orc.orc_reset_syndrome_map()
        assert lib.btbb_init(n_init) == 0 and orc.orc_init(n_init) == 0
        assert lib.btbbx_table_errors() == n_init
        for me in (2, n_init - 1, n_init, 6):
            got = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, me))
            assert got == _libs.orc_find_all(sym, n, _libs.LAP_ANY, me), (n_init, me)
        assert len(got) > 150
    finally:
        lib.btbbx_shutdown()
        orc.orc_reset_syndrome_map()
        bt.init(2)
        orc.orc_init(2)


@pytest.mark.parametrize("n_init", [3, 4])
def test_three_and_four_error_tables_two_level_kernel(n_init):
    """btbb_init(3) and btbb_init(4) run scan_slide_kernel in its two-level form (one 1024-thread workgroup per CU, complemented check stream, the
    set's members looked up in L2 a pass later).  Against the oracle with the same tables: several streams with a pitch, LSB- and
    MSB-first words, search lengths around word and tile boundaries (tiles of 1008 words here), every max_ac_errors the tables
    serve; then a stream made of sync words only (every candidate ring overflows into the in-place check) and one of sync words
    with exactly n_init errors each."""
    lib = bt.lib()
    orc = _libs.oracle()
    n_streams, nwords, pitch = 3, 3 * 1024 + 131, 3 * 1024 + 140
    try:
        lib.btbbx_shutdown()
        orc.orc_reset_syndrome_map()
        assert lib.btbb_init(n_init) == 0 and orc.orc_init(n_init) == 0
        lsb_rows, msb_rows, syms = [], [], []
        for ch in range(n_streams):
            words, inj = synth.make_stream(470 + ch, nwords, stride=512, err_cycle=7)
            sym = np.ascontiguousarray(synth.unpack_bits(words))
            msb = np.packbits(sym, bitorder="big").view(np.uint64)
            lsb_rows.append(np.concatenate([words, np.zeros(pitch - nwords, np.uint64)]))
            msb_rows.append(np.concatenate([msb, np.zeros(pitch - nwords, np.uint64)]))
            syms.append(sym)
        d_l = bt.DeviceBuffer(pitch * n_streams * 8).upload(np.concatenate(lsb_rows))
        d_m = bt.DeviceBuffer(pitch * n_streams * 8).upload(np.concatenate(msb_rows))
        cap = 1 << 16
        d_h = bt.DeviceBuffer(cap * 16)
        d_c = bt.DeviceBuffer(16)

        def run(d_w, fmt, bits, me):
            d_c.zero()
            bt.check(lib.btbbx_scan_device_fmt(d_w.ptr, nwords, pitch, n_streams, bits, bt.LAP_ANY, me, fmt, d_h.ptr, cap, d_c.ptr, None), "scan_fmt")
            bt.check(lib.btbbx_sync(None))
            cnt = int(d_c.download(np.uint32, 4)[0])
            return sorted((int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in d_h.download(bt.HIT_DTYPE, cap)[:cnt])

        total = 0
        # (a tile of this form of the kernel is 16 waves x 63 words = 1008 words)
        for bits, mes in ((nwords * 64 - 63, (0, 2, 3, 4, 6)), (nwords * 64 - 63 - 29, (n_init,)), (1024 * 64, (n_init,)), (1008 * 64 + 1, (n_init,)),
                          (1008 * 64 - 1, (n_init,)), (1008 * 64, (n_init,)), (2016 * 64 + 33, (3, 4)), (63 * 64 + 7, (n_init,)), (65, (n_init,)), (1, (n_init,))):
            for me in mes:
                want = sorted((ch, o, l, e) for ch in range(n_streams) for (o, l, e) in _libs.orc_find_all(syms[ch], bits, _libs.LAP_ANY, me))
                assert run(d_l, 0, bits, me) == want, (bits, me)
                assert run(d_m, 2, bits, me) == want, (bits, me, "msb")
                total += len(want)
        assert total > 1500
        for b in (d_l, d_m, d_h, d_c):
            b.free()
        # the drop-in entry (first match only: the kernel's atomicMin mode) with these tables
        sym, off, found, pkt = syms[0], 0, 0, C.c_void_p(None)
        while found < 12:
            n = len(sym) - 63 - off
            lo, eo = C.c_uint32(0), C.c_uint8(0)
            want = orc.orc_find_ac(C.c_void_p(sym.ctypes.data + off), n, _libs.LAP_ANY, n_init, C.byref(lo), C.byref(eo))
            got = lib.btbb_find_ac(C.c_void_p(sym.ctypes.data + off), n, bt.LAP_ANY, n_init, C.byref(pkt))
            assert (got if got >= 0 else -1) == want and want >= 0
            assert lib.btbb_packet_get_lap(pkt) == lo.value and lib.btbb_packet_get_ac_errors(pkt) == eo.value
            found += 1
            off += want + 1
        rng = np.random.default_rng(_libs.seed(44))
        laps = rng.integers(0, 1 << 24, 4096)
        clean = [synth.syncword(int(l)) for l in laps]
        hurt = [w ^ sum(1 << int(b) for b in rng.choice(57, n_init, replace=False)) for w in clean]  # n_init errors below the barker bits
        for body in (clean, hurt):
            words = np.array(body + [0], dtype=np.uint64)
            sym = np.ascontiguousarray(synth.unpack_bits(words))
            n = len(sym) - 63
            for me in (n_init - 1, n_init):
                got = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, me))
                assert got == _libs.orc_find_all(sym, n, _libs.LAP_ANY, me)
            assert len(got) >= 4096
    finally:
        lib.btbbx_shutdown()
        orc.orc_reset_syndrome_map()
        bt.init(2)
        orc.orc_init(2)


def test_find_ac_drop_in():
    """btbb_find_ac through the C ABI: first match, packet allocation, LAP / ac_errors."""
    lib = bt.lib()
    orc = _libs.oracle()
    words, sym, _ = stream(106, 1 << 10, stride=2048)
    for lap in (bt.LAP_ANY, 0x9E8B33):
        off = 0
        found = 0
        pkt = C.c_void_p(None)
        while True:
            n = len(sym) - 63 - off
            if n <= 0:
                break
            lo, eo = C.c_uint32(0), C.c_uint8(0)
            want = orc.orc_find_ac(C.c_void_p(sym.ctypes.data + off), n, lap, 2, C.byref(lo), C.byref(eo))
            got = lib.btbb_find_ac(C.c_void_p(sym.ctypes.data + off), n, lap, 2, C.byref(pkt))
            assert (got if got >= 0 else -1) == want
            if want < 0:
                break
            assert lib.btbb_packet_get_lap(pkt) == lo.value and lib.btbb_packet_get_ac_errors(pkt) == eo.value
            assert lib.btbb_packet_get_flag(pkt, 0) == 1
            found += 1
            off += want + 1
            if lap != bt.LAP_ANY:
                break
        if lap == bt.LAP_ANY:
            assert found > 20
        if pkt.value:
            lib.btbb_packet_unref(pkt)
    # nothing found: *pkt stays NULL
    z = np.zeros(5000, np.uint8)
    pkt = C.c_void_p(None)
    assert lib.btbb_find_ac(_libs.ptr(z), 4000, bt.LAP_ANY, 2, C.byref(pkt)) < 0 and not pkt.value


def test_find_ac_walk_is_answered_from_the_remembered_window_and_only_while_it_is_true():
    """A caller that walks a buffer with btbb_find_ac is answered, from its second call on, from the list of ALL matches of
    the window scanned then (btbb_api.cpp, AcWindow) -- as long as the bytes are the bytes that were scanned.  Every call
    of several walks is held against the oracle: a plain walk, a walk whose buffer is changed between two calls (an access
    code destroyed behind the cursor, one created in front of it), alternating search parameters, windows that end
    elsewhere, a buffer with more matches than the list holds, and one with none."""
    lib = bt.lib()
    orc = _libs.oracle()
    rng = np.random.default_rng(_libs.seed(141))

    def call(sym, off, n, lap, max_err, pkt):
        lo, eo = C.c_uint32(0), C.c_uint8(0)
        want = orc.orc_find_ac(C.c_void_p(sym.ctypes.data + off), n, lap, max_err, C.byref(lo), C.byref(eo))
        got = lib.btbb_find_ac(C.c_void_p(sym.ctypes.data + off), n, lap, max_err, C.byref(pkt))
        assert (got if got >= 0 else -1) == want, (off, n, hex(lap), max_err, got, want)
        if want >= 0:
            assert lib.btbb_packet_get_lap(pkt) == lo.value and lib.btbb_packet_get_ac_errors(pkt) == eo.value
        return want

    def walk(sym, lap, max_err, pkt, mutate=None):
        off, found = 0, 0
        while True:
            n = len(sym) - 63 - off
            if n <= 0:
                break
            r = call(sym, off, n, lap, max_err, pkt)
            if r < 0:
                break
            found += 1
            off += r + 1
            if mutate is not None and found == mutate[0]:
                mutate[1](off)
        return found

    _, sym, _ = stream(141, 1 << 9, stride=1024)               # 32 Ki symbols, an access code every 1024
    sym = np.ascontiguousarray(sym)
    pkt = C.c_void_p(None)
    n_all = walk(sym, bt.LAP_ANY, 2, pkt)
    assert n_all > 20
    # the same buffer again (remembered), then with one access code in front of the cursor broken and a new one planted
    assert walk(sym, bt.LAP_ANY, 2, pkt) == n_all
    planted = synth.access_code(0x2A5B17)[:64]

    def change(off):
        nxt = orc.orc_find_ac(C.c_void_p(sym.ctypes.data + off), len(sym) - 63 - off, bt.LAP_ANY, 2, C.byref(C.c_uint32(0)), C.byref(C.c_uint8(0)))
        assert nxt >= 0
        sym[off + nxt + 10:off + nxt + 20] ^= 1                 # the next match is gone
        sym[off + nxt + 200:off + nxt + 264] = planted          # and a new one stands 200 symbols behind where it was
    m = walk(sym, bt.LAP_ANY, 2, pkt, mutate=(5, change))
    assert m in (n_all, n_all + 1)
    # (round 5: a call compares only the symbols its answer rests on -- up to the end of the match it returns.)  A change
    # FAR in front of the cursor, behind several matches still to come, must be seen when the walk gets there
    def change_far(off):
        p = off
        for _ in range(4):
            nxt = orc.orc_find_ac(C.c_void_p(sym.ctypes.data + p), len(sym) - 63 - p, bt.LAP_ANY, 2, C.byref(C.c_uint32(0)), C.byref(C.c_uint8(0)))
            assert nxt >= 0
            p += nxt + 1
        sym[p - 1 + 5:p - 1 + 25] ^= 1                          # the fourth match from here is gone
        sym[p + 300:p + 364] = synth.access_code(0x5C3A91)[:64]  # and another one stands behind it
    m2 = walk(sym, bt.LAP_ANY, 2, pkt, mutate=(3, change_far))
    assert abs(m2 - m) <= 1
    # parameters alternate call by call on one buffer; windows with other ends
    for k in range(12):
        off = int(rng.integers(0, len(sym) - 3000))
        call(sym, off, len(sym) - 63 - off, bt.LAP_ANY if k % 2 else 0x2A5B17, 1 + k % 2, pkt)
        call(sym, off, int(rng.integers(1, 2500)), bt.LAP_ANY, 2, pkt)
    # more matches than the list holds: back-to-back sync words (4 500 of them), walked to the end
    dense = np.concatenate([synth.access_code(int(l))[:64] for l in rng.integers(0, 1 << 24, 4500)]).astype(np.uint8)
    dense = np.ascontiguousarray(np.concatenate([dense, np.zeros(70, np.uint8)]))
    assert walk(dense, bt.LAP_ANY, 0, pkt) >= 4500
    # and none at all
    z = np.zeros(9000, np.uint8)
    for off in (0, 10, 20):
        assert call(z, off, len(z) - 63 - off, bt.LAP_ANY, 2, pkt) < 0
    if pkt.value:
        lib.btbb_packet_unref(pkt)


def test_large_stream_properties():
    """512 MiB of packed stream generated in HBM: every injected sync word with <= 2 bit errors
    is reported with its LAP and error count, nothing is reported twice, the count equals
    injections + a plausible number of chance matches, and two sampled slices equal the oracle."""
    import torch
    lib = bt.lib()
    nwords = 1 << 26                       # 2^32 symbols
    stride = 4096
    t = torch.empty(nwords + 8, dtype=torch.int64, device="cuda")
    bt.check(lib.btbbx_synth_device(t.data_ptr(), 0, nwords, 7, stride, -1, 4, None))
    cap = 1 << 21
    hits_t = torch.zeros(cap * 2, dtype=torch.int64, device="cuda")
    cnt_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    nbits = nwords * 64 - 63
    bt.check(lib.btbbx_scan_device(t.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2,
                                   hits_t.data_ptr(), cap, cnt_t.data_ptr(), None))
    torch.cuda.synchronize()
    cnt = int(cnt_t.item())
    assert cnt <= cap
    hits = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:cnt]
    hits = hits[np.argsort(hits["offset"], kind="stable")]
    assert len(np.unique(hits["offset"])) == cnt
    k = np.arange(nbits // stride, dtype=np.uint64)
    pos, laps, nerr, mask = synth.injection_params(7, k, stride, 4)
    popc = np.unpackbits(mask.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1)   # repeated positions cancel
    ok = (popc <= 2) & (pos + np.uint64(64) <= np.uint64(nwords * 64))
    idx = np.searchsorted(hits["offset"], pos[ok])
    assert (idx < cnt).all() and (hits["offset"][idx] == pos[ok]).all()
    assert (hits["lap"][idx] == laps[ok]).all()
    assert (hits["ac_errors"][idx] == popc[ok]).all()
    extra = cnt - int(ok.sum())
    assert 0 <= extra < 400                # theory: ~1.25e-8 * 2^32 = 54 chance matches
    # sampled slices against the oracle
    for first in (12345, nwords - 40000):
        nw = 32768
        sl = t[first:first + nw].cpu().numpy().view(np.uint64)
        sym = np.ascontiguousarray(synth.unpack_bits(sl))
        want = _libs.orc_find_all(sym, nw * 64 - 63, _libs.LAP_ANY, 2)
        lo, hi = first * 64, first * 64 + nw * 64 - 63
        sel = hits[(hits["offset"] >= lo) & (hits["offset"] < hi)]
        got = [(int(h["offset"]) - lo, int(h["lap"]), int(h["ac_errors"])) for h in sel]
        assert got == want


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_streaming_ingest(fmt):
    """Chunked feed (pinned double buffers, one-word carry): same hits as one scan of the whole
    capture, access codes straddling chunk boundaries found exactly once, ragged last chunk."""
    lib = bt.lib()
    words, sym, _ = stream(107, 3000, stride=512)
    total = len(sym) - 37                      # ragged end
    want = _libs.orc_find_all(sym, total - 63, _libs.LAP_ANY, 2)
    for chunk in (64, 640, 4096, 65536):
        h = lib.btbbx_stream_open(bt.LAP_ANY, 2, chunk, fmt)
        assert h, lib.btbbx_last_error()
        got = []
        buf = np.zeros(1 << 16, bt.HIT_DTYPE)
        pos = 0
        while pos < total:
            n = min(chunk, total - pos)
            if fmt == 1:
                data = np.ascontiguousarray(sym[pos:pos + n])
            elif fmt == 0:
                data = np.ascontiguousarray(words[pos // 64:(pos + n + 63) // 64])
            else:       # 8 symbols per byte, MSB first
                pad = np.concatenate([sym[pos:pos + n], np.zeros((-n) % 64, np.uint8)])
                data = np.ascontiguousarray(np.packbits(pad, bitorder="big"))
            k = bt.check(lib.btbbx_stream_feed(h, data.ctypes.data, n, buf.ctypes.data, len(buf)), "feed")
            got += as_tuples(buf[:k])
            pos += n
        k = bt.check(lib.btbbx_stream_flush(h, buf.ctypes.data, len(buf)), "flush")
        got += as_tuples(buf[:k])
        lib.btbbx_stream_close(h)
        assert got == want, (fmt, chunk, len(got), len(want))
    assert len(want) > 100


def test_device_hit_sort():
    """btbbx_sort_hits_device orders (stream, offset) like the host routine, keeps every record."""
    lib = bt.lib()
    rng = np.random.default_rng(7)
    # (the sort key holds only as many offset and stream bits as the list needs: 1 .. 64 of them)
    for n, streams, maxoff in ((1, 1, 10), (2, 2, 100), (777, 3, 1 << 20), (100000, 79, 1 << 33), (2500000, 1, 1 << 40),
                               (3, 1, 1), (1000, 40000, 1 << 4), (5000, 65536, 1 << 47)):
        h = np.zeros(n, bt.HIT_DTYPE)
        h["offset"] = rng.integers(0, maxoff, n, dtype=np.uint64)
        h["stream"] = rng.integers(0, streams, n)
        h["lap"] = rng.integers(0, 1 << 24, n)
        h["ac_errors"] = rng.integers(0, 6, n)
        d = bt.DeviceBuffer(h.nbytes).upload(h)
        bt.check(lib.btbbx_sort_hits_device(d.ptr, n, None))
        got = d.download(bt.HIT_DTYPE, n)
        d.free()
        host = h.copy()
        lib.btbbx_sort_hits(host.ctypes.data, n)
        key = lambda a: (a["stream"].astype(np.uint64) << np.uint64(48)) | a["offset"]
        assert np.array_equal(key(got), key(host)) and np.all(np.diff(key(got).astype(np.int64)) >= 0)
        full = ["stream", "offset", "lap", "ac_errors"]
        assert np.array_equal(np.sort(got, order=full), np.sort(h, order=full))      # a permutation of the input


def test_device_hit_order_with_the_count_in_hbm():
    """btbbx_order_hits_device: the list's length is read from device memory (the scan's own counter), the scratch
    is the caller's, nothing is synchronised in between.  Cases: sparse multi-stream lists, a crowded bucket (runs of
    consecutive offsets: the presence-bitmap path), a short list over a huge key space with more than 48 records in
    one bucket (the all-pairs path), repeated keys, a count above / below the capacity, empty and one-record lists."""
    lib = bt.lib()
    rng = np.random.default_rng(_libs.seed(11))

    def run(h, cap, count=None, bounded=0):
        n = len(h)
        count = n if count is None else count
        room = np.zeros(max(cap, 1), bt.HIT_DTYPE)
        room[:min(n, cap)] = h[:cap]
        d = bt.DeviceBuffer(room.nbytes).upload(room)
        c = bt.DeviceBuffer(16).upload(np.array([count, 0, 0, 0], np.uint32))
        sb = lib.btbbx_order_hits_scratch_bytes(cap)
        scratch = bt.DeviceBuffer(sb)
        if bounded:                                                   # the form for a scan's own list: bounds from the caller
            bt.check(lib.btbbx_order_scan_hits_device(d.ptr, c.ptr, cap, int(room["stream"].max()) + 1 + bounded - 1,
                                                      int(room["offset"].max()) + bounded, scratch.ptr, sb, None), "order_scan_hits")
        else:
            bt.check(lib.btbbx_order_hits_device(d.ptr, c.ptr, cap, scratch.ptr, sb, None), "btbbx_order_hits_device")
        bt.check(lib.btbbx_sync(None))
        got = d.download(bt.HIT_DTYPE, max(cap, 1))
        for b in (d, c, scratch):
            b.free()
        m = min(count, cap)
        want = np.sort(room[:m], order=["stream", "offset"], kind="stable")
        key = lambda a: (a["stream"].astype(np.uint64) << np.uint64(48)) | a["offset"]
        assert np.array_equal(key(got[:m]), key(want)), (n, cap, count)
        full = ["stream", "offset", "lap", "ac_errors"]
        assert np.array_equal(np.sort(got[:m], order=full), np.sort(room[:m], order=full))
        assert np.array_equal(got[m:], room[m:])                       # records behind the count are left alone

    def hits(offsets, streams):
        h = np.zeros(len(offsets), bt.HIT_DTYPE)
        h["offset"], h["stream"] = offsets, streams
        h["lap"] = rng.integers(0, 1 << 24, len(h))
        h["ac_errors"] = rng.integers(0, 3, len(h))
        return h[rng.permutation(len(h))]

    # sparse: 79 streams, one hit per ~4096 offsets of 2^26
    off = np.concatenate([np.sort(rng.choice(1 << 26, 16000, replace=False)) for _ in range(79)]).astype(np.uint64)
    st = np.repeat(np.arange(79), 16000)
    run(hits(off, st), cap=79 * 16000 + 5000)
    run(hits(off, st), cap=79 * 16000 + 5000, count=1000)              # a short list in a big buffer
    run(hits(off, st), cap=79 * 16000 + 5000, bounded=1)               # bounds given: exactly the extent ...
    run(hits(off, st), cap=79 * 16000 + 5000, bounded=12345)           # ... and loose ones
    run(hits(off, st)[:3000], cap=2000, count=3000)                    # the counter ran past the capacity
    # a list of more than 4 M records (round 5: up to 2^24 buckets, sixteen counters per thread in the scans of the counts):
    # 6 M hits of one stream, one per ~5 400 offsets of 2^35 as in the headline's list, with and without bounds
    big = np.sort(rng.choice(1 << 35, 6_000_000, replace=False)).astype(np.uint64)
    run(hits(big, 0), cap=len(big) + 70_000)
    run(hits(big, 0), cap=len(big) + 70_000, bounded=1)
    # crowded: 300 000 consecutive offsets (a stream made of sync words) next to sparse ones
    off = np.concatenate([np.arange(5_000_000, 5_300_000), rng.choice(1 << 33, 50000, replace=False)]).astype(np.uint64)
    run(hits(np.unique(off), 0), cap=400000)
    # few buckets, huge key space, 200 records inside one bucket: all pairs
    off = np.concatenate([(1 << 40) + np.arange(0, 200 * 977, 977), rng.choice(1 << 41, 100, replace=False)]).astype(np.uint64)
    run(hits(np.unique(off), 0), cap=400)
    # repeated keys (no scan produces them)
    off = rng.integers(0, 50, 5000).astype(np.uint64)
    run(hits(off, rng.integers(0, 3, 5000)), cap=5000)
    off = np.concatenate([np.repeat(np.arange(100000, 100100), 3), np.arange(100100, 160000), [1 << 33]]).astype(np.uint64)
    run(hits(off, 0), cap=70000)                                      # a crowded bucket (bitmap path) with repeats
    for n in (0, 1, 2):
        run(hits(np.arange(n, dtype=np.uint64) * 77, 0), cap=max(n, 4))


@pytest.mark.parametrize("slots", [False, True])
@pytest.mark.parametrize("lap", [bt.LAP_ANY, 0x9E8B33, 0x1E8B33])
def test_scan_ordered_device(lap, slots):
    """btbbx_scan_ordered_device: one call, the list comes back in (stream, offset) order and equals the oracle's per
    stream -- LAP_ANY and known LAPs of both barker classes, several streams with a pitch, a hit buffer smaller than the
    number of matches (the records kept are then some subset, still ordered, and the counter says how many there were).
    slots: the scratch of btbbx_scan_ordered_scratch_bytes -- the scan then leaves its hits in the segment slots (round 6; a sync
    word every 512 symbols = eight hits per segment of two slots: most of the list goes through the overflow list); else
    btbbx_order_hits_scratch_bytes -- the general ordering."""
    lib = bt.lib()
    kw = dict(stride=512) if lap == bt.LAP_ANY else dict(stride=512, lap=lap)
    n_streams, nwords, pitch = 5, 3000 + 7, 3100
    rows, want = [], []
    for ch in range(n_streams):
        words, sym, _ = stream(140 + ch, nwords, **kw)
        rows.append(np.concatenate([words, np.zeros(pitch - nwords, np.uint64)]))
        n = len(sym) - 63 - 11
        want += [(ch, o, l, e) for (o, l, e) in _libs.orc_find_all(sym, n, lap if lap != bt.LAP_ANY else _libs.LAP_ANY, 2)]
    buf = np.concatenate(rows)
    d_w = bt.DeviceBuffer(buf.nbytes).upload(buf)
    # (capacities around 2^21 and 2^22: the bucket count of the ordering -- one more bit when the caller gives the bounds -- is at
    # its largest there)
    # (... and beyond 2^22: round 5 lets lists of more than 4 M records have up to 2^24 buckets, sixteen counters per thread in the
    # scans of the counts)
    for cap in (len(want) + 100, len(want) // 3, (1 << 21) - 3, (1 << 21) + 1, (1 << 22) + 5, (1 << 23) + 5, (1 << 24) + 3):
        d_h = bt.DeviceBuffer(cap * 16).zero()
        d_c = bt.DeviceBuffer(16).zero()
        sb = lib.btbbx_scan_ordered_scratch_bytes(nwords * 64 - 63 - 11, n_streams, lap, cap) if slots else lib.btbbx_order_hits_scratch_bytes(cap)
        assert (sb > lib.btbbx_order_hits_scratch_bytes(cap)) == slots
        d_s = bt.DeviceBuffer(sb)
        bt.check(lib.btbbx_scan_ordered_device(d_w.ptr, nwords, pitch, n_streams, nwords * 64 - 63 - 11, lap, 2, d_h.ptr, cap, d_c.ptr,
                                               d_s.ptr, sb, None), "btbbx_scan_ordered_device")
        bt.check(lib.btbbx_sync(None))
        cnt = int(d_c.download(np.uint32, 4)[0])
        got = d_h.download(bt.HIT_DTYPE, cap)[:min(cnt, cap)]
        for b in (d_h, d_c, d_s):
            b.free()
        assert cnt == len(want) > 500
        tup = [(int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in got]
        assert tup == sorted(tup)
        if cap >= cnt:
            assert tup == want
        else:
            # (segment slots: the compaction writes positions 0 .. cap - 1 of the whole list -- the cap smallest records --
            # unless the overflow list, cap entries, ran full as well and the general ordering redid the call: eight hits per
            # two-slot segment here)
            assert len(tup) == cap and set(tup) <= set(want)
    d_w.free()


@pytest.mark.parametrize("lap", [bt.LAP_ANY, 0x9E8B33, 0x1E8B33])
def test_msb_first_capture_scanned_as_it_is(lap):
    """BTBBX_FMT_PACKED_MSB (8 symbols per byte, first symbol in bit 7 -- a dongle's dump): the scan kernels turn the dwords
    round in registers, the capture in HBM is not rewritten.  Hit for hit what the LSB-first words of the same symbols give
    and what the oracle finds, for several streams with a pitch, ragged search lengths around word and tile boundaries, through
    btbbx_scan_device_fmt and btbbx_scan_ordered_device_fmt (the latter entered with a stale counter: it zeroes it itself)."""
    lib = bt.lib()
    kw = dict(stride=512) if lap == bt.LAP_ANY else dict(stride=512, lap=lap)
    olap = lap if lap != bt.LAP_ANY else _libs.LAP_ANY
    n_streams, nwords, pitch = 3, 2 * 768 + 131, 2 * 768 + 140
    lsb_rows, msb_rows, syms = [], [], []
    for ch in range(n_streams):
        words, sym, _ = stream(170 + ch, nwords, **kw)
        msb = np.packbits(sym, bitorder="big").view(np.uint64)          # the same symbols, MSB first in every byte
        assert len(msb) == nwords and not np.array_equal(msb, words)
        lsb_rows.append(np.concatenate([words, np.zeros(pitch - nwords, np.uint64)]))
        msb_rows.append(np.concatenate([msb, np.zeros(pitch - nwords, np.uint64)]))
        syms.append(sym)
    d_l = bt.DeviceBuffer(pitch * n_streams * 8).upload(np.concatenate(lsb_rows))
    d_m = bt.DeviceBuffer(pitch * n_streams * 8).upload(np.concatenate(msb_rows))
    cap = 1 << 16
    d_h = bt.DeviceBuffer(cap * 16)
    d_c = bt.DeviceBuffer(16)
    sb = lib.btbbx_order_hits_scratch_bytes(cap)
    sb_slots = lib.btbbx_scan_ordered_scratch_bytes(nwords * 64 - 63, n_streams, lap, cap)    # (LAP_ANY: with the segment slots)
    d_s = bt.DeviceBuffer(max(sb, sb_slots))

    def plain(d_w, fmt, bits):
        d_c.zero()
        bt.check(lib.btbbx_scan_device_fmt(d_w.ptr, nwords, pitch, n_streams, bits, lap, 2, fmt, d_h.ptr, cap, d_c.ptr, None), "scan_fmt")
        bt.check(lib.btbbx_sync(None))
        cnt = int(d_c.download(np.uint32, 4)[0])
        return sorted((int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in d_h.download(bt.HIT_DTYPE, cap)[:cnt])

    def ordered(d_w, fmt, bits, scratch_bytes=None):
        d_c.upload(np.array([12345, 0, 0, 0], np.uint32))               # stale: the call must not trust it
        bt.check(lib.btbbx_scan_ordered_device_fmt(d_w.ptr, nwords, pitch, n_streams, bits, lap, 2, fmt, d_h.ptr, cap, d_c.ptr,
                                                   d_s.ptr, scratch_bytes or sb, None), "scan_ordered_fmt")
        bt.check(lib.btbbx_sync(None))
        cnt = int(d_c.download(np.uint32, 4)[0])
        return [(int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in d_h.download(bt.HIT_DTYPE, cap)[:cnt]]

    total = 0
    # (a tile of the LAP_ANY kernel is 12 waves x 63 words = 756 words, of the known-LAP kernel 512)
    for bits in (nwords * 64 - 63, nwords * 64 - 63 - 29, 768 * 64, 768 * 64 + 1, 768 * 64 - 1, 756 * 64, 756 * 64 + 1, 756 * 64 - 1,
                 2 * 756 * 64 - 63, 63 * 64 + 5, 64 * 700 + 33, 65, 1):
        want = sorted((ch, o, l, e) for ch in range(n_streams) for (o, l, e) in _libs.orc_find_all(syms[ch], bits, olap, 2))
        got_m = plain(d_m, 2, bits)
        assert got_m == plain(d_l, 0, bits) == want, (lap, bits, len(got_m), len(want))
        assert ordered(d_m, 2, bits) == want == ordered(d_l, 0, bits), (lap, bits)
        assert ordered(d_m, 2, bits, sb_slots) == want == ordered(d_l, 0, bits, sb_slots), (lap, bits, "segment slots")
        total += len(want)
    assert total > 300
    with pytest.raises(bt.BtbbError):
        bt.check(lib.btbbx_scan_device_fmt(d_m.ptr, nwords, pitch, n_streams, 64, lap, 2, 1, d_h.ptr, cap, d_c.ptr, None), "scan_fmt")
    for b in (d_l, d_m, d_h, d_c, d_s):
        b.free()


@pytest.mark.parametrize("lap", [bt.LAP_ANY, 0x9E8B33])
def test_sharded_product_scan_equals_single_scan(lap):
    """btbbx_scan_host_multi (the C-ABI form of the N-GPU path): the library's own shard plan --
    word-aligned slices + 63-symbol halo -- run through the PRODUCT kernels, one host thread per
    listed device, equals the single scan and the oracle.  Here every shard runs on device 0."""
    kw = dict(stride=512) if lap == bt.LAP_ANY else dict(stride=512, lap=lap)
    words, sym, _ = stream(131, 3000 + 7, **kw)
    n = len(sym) - 63 - 11                              # not word aligned on purpose
    want = _libs.orc_find_all(sym, n, lap if lap != bt.LAP_ANY else _libs.LAP_ANY, 2)
    assert as_tuples(bt.scan_words(words, n, lap, 2)) == want and len(want) > 250
    bt.init_devices([0], 2)
    for shards in (1, 2, 3, 8):
        got = bt.scan_words_multi(words, n, [0] * shards, lap, 2)
        assert as_tuples(got) == want, shards
        # ... and shard by shard, the way one-process-per-GPU ranks use the plan (bench.py --gpus N)
        parts, pos = [], 0
        for k in range(shards):
            p = bt.shard_plan(n, shards, k)
            assert p["first_offset"] == pos and p["first_offset"] == 64 * p["first_word"]
            pos += p["search_bits"]
            if p["search_bits"]:
                assert p["search_bits"] + 63 <= 64 * p["n_words"] and p["first_word"] + p["n_words"] <= len(words)
                h = bt.scan_words(words[p["first_word"]:p["first_word"] + p["n_words"]], p["search_bits"], lap, 2)
                parts += [(int(x["offset"]) + p["first_offset"], int(x["lap"]), int(x["ac_errors"])) for x in h]
        assert pos == n and parts == want, shards
    with pytest.raises(bt.BtbbError):
        bt.scan_words_multi(words, n, [0, 99], lap, 2)              # no such device


def test_truncated_host_scan_keeps_the_smallest_hits():
    """`cap` smaller than the number of matches: the host wrappers hand back the cap SMALLEST
    (stream, offset) hits -- cap = 1 is btbb_find_ac's first match -- not whichever wavefront won."""
    words, sym, _ = stream(132, 1 << 13, stride=512)                 # ~1000 hits
    n = len(sym) - 63
    full = as_tuples(bt.scan_words(words, n, bt.LAP_ANY, 2))
    assert len(full) > 700
    for cap in (1, 2, 7, 64, 500):
        for _ in range(3):                                           # the race is different every time
            assert as_tuples(bt.scan_words(words, n, bt.LAP_ANY, 2, cap=cap, truncate=True)) == full[:cap], cap
        assert as_tuples(bt.scan_words_multi(words, n, [0, 0, 0], bt.LAP_ANY, 2, cap=cap, truncate=True)) == full[:cap]
    hits = np.zeros(4, dtype=bt.HIT_DTYPE)
    s8 = np.ascontiguousarray(sym)
    cnt = bt.lib().btbbx_scan_symbols(_libs.ptr(s8), len(s8), n, bt.LAP_ANY, 2, _libs.ptr(hits), 4)
    assert cnt == len(full) and as_tuples(hits) == full[:4]


def test_concurrent_drop_in_callers():
    """Several host threads inside btbb_find_ac / btbbx_scan_host at once, each on its own buffer
    (the reference's functions only touch the caller's data, so multi-channel callers do this):
    every call leases private scratch memory and a private stream, results equal the sequential ones."""
    import threading
    lib = bt.lib()
    jobs = []
    for k in range(6):
        words, sym, _ = stream(140 + k, 512 + 64 * k, stride=2048 + 512 * k)
        jobs.append((words, np.ascontiguousarray(sym), len(sym) - 63))
    want_scan = [as_tuples(bt.scan_words(w, n, bt.LAP_ANY, 2)) for w, s, n in jobs]
    want_first = []
    for w, s, n in jobs:
        pkt = C.c_void_p(None)
        want_first.append(lib.btbb_find_ac(_libs.ptr(s), min(n, 60000), bt.LAP_ANY, 2, C.byref(pkt)))
        if pkt.value:
            lib.btbb_packet_unref(pkt)
    errors = []

    def worker(k):
        try:
            w, s, n = jobs[k]
            for rep in range(25):
                assert as_tuples(bt.scan_words(w, n, bt.LAP_ANY, 2)) == want_scan[k]
                pkt = C.c_void_p(None)
                assert lib.btbb_find_ac(_libs.ptr(s), min(n, 60000), bt.LAP_ANY, 2, C.byref(pkt)) == want_first[k]
                if pkt.value:
                    assert want_scan[k] and lib.btbb_packet_get_lap(pkt) == want_scan[k][0][1]
                    lib.btbb_packet_unref(pkt)
        except Exception as e:                                       # noqa: BLE001 -- reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("lap", [bt.LAP_ANY, 0x9E8B33, 0x1E8B33])
def test_hits_at_the_seams_of_lanes_segments_waves_and_tiles(lap):
    """Sync words planted ON the boundaries of what the scan kernels hand out -- scan_known_lap_kernel (round 6): a lane's run of two
    words (128 offsets), a 4096-offset segment, a wave's 128 words, a tile of 512 words; scan_slide_kernel: a word, a wave's 63 words,
    a tile of 756 -- at the offsets right before, on and behind each boundary and with windows that straddle it, 0 .. 2 errors each,
    both barker classes, in fourteen streams with a pitch (every boundary meets every delta); the plain and the ordered call (segment slots) against the oracle."""
    lib = bt.lib()
    olap = _libs.LAP_ANY if lap == bt.LAP_ANY else lap
    rng = np.random.default_rng(77)
    deltas = (-65, -64, -63, -33, -32, -31, -1, 0, 1, 31, 32, 33, 63, 64)
    bounds = (1, 2, 63, 64, 126, 128, 189, 256, 512, 756, 1024, 1512)                 # in words
    n_streams, nwords, pitch = len(deltas), 2 * 512 + 2 * 756 + 37, 2 * 512 + 2 * 756 + 64
    nbits = nwords * 64 - 63 - 5
    rows, want, planted = [], [], 0
    for ch in range(n_streams):
        words = synth.noise_words(900 + ch, 0, nwords)
        sym = np.ascontiguousarray(synth.unpack_bits(words))
        last = -1000
        for j, b in enumerate(bounds):                          # stream ch: boundary j gets the delta (ch + 3 j) mod 14
            off = b * 64 + deltas[(ch + 3 * j) % len(deltas)]
            if off < 0 or off + 64 > len(sym) or off - last < 64 + 7:
                continue
            sw = synth.syncword(lap if lap != bt.LAP_ANY else int(rng.integers(0, 1 << 24)))
            for e in rng.choice(57, size=(j + ch) % 3, replace=False):           # errors below the barker bits
                sw ^= 1 << int(e)
            sym[off:off + 64] = synth.bits_lsb(sw, 64)
            last = off
            planted += 1
        words = synth.pack_bits(sym)
        rows.append(np.concatenate([words, np.zeros(pitch - nwords, np.uint64)]))
        want += [(ch, o, l, e) for (o, l, e) in _libs.orc_find_all(sym, nbits, olap, 2)]
    assert planted >= 150 and len(want) >= planted
    buf = np.concatenate(rows)
    d_w = bt.DeviceBuffer(buf.nbytes).upload(buf)
    cap = len(want) + 50
    for ordered in (False, True):
        d_h = bt.DeviceBuffer(cap * 16).zero()
        d_c = bt.DeviceBuffer(16).zero()
        if ordered:
            sb = lib.btbbx_scan_ordered_scratch_bytes(nbits, n_streams, lap, cap)
            d_s = bt.DeviceBuffer(sb)
            bt.check(lib.btbbx_scan_ordered_device(d_w.ptr, nwords, pitch, n_streams, nbits, lap, 2, d_h.ptr, cap, d_c.ptr, d_s.ptr, sb, None))
        else:
            bt.check(lib.btbbx_scan_device(d_w.ptr, nwords, pitch, n_streams, nbits, lap, 2, d_h.ptr, cap, d_c.ptr, None))
        bt.check(lib.btbbx_sync(None))
        cnt = int(d_c.download(np.uint32, 4)[0])
        got = d_h.download(bt.HIT_DTYPE, cap)[:min(cnt, cap)]
        tup = [(int(h["stream"]), int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in got]
        if ordered:
            assert tup == want
            d_s.free()
        else:
            assert sorted(tup) == want
        d_h.free()
        d_c.free()
    d_w.free()
