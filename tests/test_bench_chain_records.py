"""bench.py's in-run parity of the config-3 lines compares the GPU's btbbx_pkt_out records with what the UNMODIFIED
reference decodes per access code (oracle/ref_internals.c refint_known_lap_chain_records).  Here, without a GPU: that
helper against the counting form the round-3 bench used, against the oracle port packet by packet, and bench.py's
numpy restatement of its payload hash (computed there from btbbx_pkt_out words) against the hash the C side forms."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _libs
import libbtbb_amd as bt
from libbtbb_amd import synth
from test_gpu_packets import _oracle_decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_chain_records_match_the_oracle_and_the_hash_formula():
    ref = _libs.ref()
    if ref is None:
        pytest.skip("compiled reference not available")
    bench = _bench()
    ref.btbb_init(2)
    orc = _libs.oracle()
    orc.orc_init(2)
    lap, uap = 0x9E8B33, 0x47
    rng = np.random.default_rng(_libs.seed(97))
    n_words = 1 << 13
    sym = synth.unpack_bits(synth.noise_words(5, 0, n_words))
    types = [synth.TYPE_DM1, synth.TYPE_DH1, synth.TYPE_DM3, synth.TYPE_DH3, synth.TYPE_DM5, synth.TYPE_DH5, synth.TYPE_FHS]
    maxbody = {synth.TYPE_DM1: 17, synth.TYPE_DH1: 27, synth.TYPE_DM3: 121, synth.TYPE_DH3: 183, synth.TYPE_DM5: 224, synth.TYPE_DH5: 339}
    slots = n_words * 64 // 4096 - 1
    for k in range(slots):
        t = types[k % 7]
        body = rng.integers(0, 256, maxbody.get(t, 0) if k % 2 else int(rng.integers(0, maxbody.get(t, 0) + 1)), dtype=np.uint8).tobytes()
        p = synth.build_packet(lap, uap, k & 63, t, lt_addr=1 + k % 7, flags=k % 8, body=body,
                               fhs_bits=synth.fhs_payload(lap, uap, 0x1234, k, rng)).copy()
        if k % 5 == 0:
            p[rng.integers(126, len(p), 3)] ^= 1                 # FEC 2/3 corrects or fails, DH CRCs fail
        if k % 9 == 0:
            p[rng.integers(68, 122, 4)] ^= 1                     # header trouble
        pos = k * 4096 + 100 + int(rng.integers(0, 64))
        sym[pos:pos + len(p)] = p
    sym = np.ascontiguousarray(sym)
    rec = np.zeros(slots + 64, bench.CHAIN_REC)
    n = int(ref.refint_known_lap_chain_records(_libs.ptr(sym), len(sym), lap, 2, uap, 4096, rec.ctypes.data_as(C.c_void_p), len(rec)))
    good = C.c_uint64(0)
    assert n == int(ref.refint_known_lap_chain(_libs.ptr(sym), len(sym), lap, 2, uap, 4096, C.byref(good)))
    rec = rec[:n]
    assert int(good.value) == int(np.isin(rec["payload_rv"], (10, 1000)).sum()) and n >= slots - 2
    # the oracle port on every match, and the hash from a btbbx_pkt_out-shaped record built from the oracle's payload bits
    out = np.zeros(n, bt.PKTOUT_DTYPE)
    seen = set()
    for i, q in enumerate(rec):
        at = int(q["offset"])
        s = np.ascontiguousarray(sym[at:at + 3125])
        present, h, r, st = _oracle_decode(orc, s, (at // 4096) & 63, uap)
        assert (int(q["header_present"]), int(q["header_rv"])) == (present, h), (i, at)
        out[i]["header_rv"] = h
        if h:
            assert int(q["payload_rv"]) == r and int(q["payload_length"]) == st["payload_length"], (i, at, r)
            assert (int(q["type"]), int(q["lt_addr"]), int(q["hdr_flags"]), int(q["hec"])) == \
                   (st["packet_type"], st["packet_lt_addr"], st["packet_flags"], st["packet_hec"]), (i, at)
            out[i]["payload_rv"], out[i]["payload_length"] = r, st["payload_length"]
            bits = np.zeros(43 * 64, np.uint8)
            bits[:2744] = st["payload"]
            out[i]["payload"] = synth.pack_bits(bits)
            seen.add((st["packet_type"], r))
    assert np.array_equal(bench.payload_hash(out), rec["payload_hash"])
    assert {(15, 10), (14, 10), (11, 10), (10, 10), (2, 1000)} <= seen and any(r == 2 for _, r in seen), seen
