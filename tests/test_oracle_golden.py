"""Oracle (oracle/btbb_oracle.c) against the reference's own known-answer vectors
(tests/golden/reference_vectors.json: tests/test_syndromes.c, tests/test_fec23.c,
tests/test_header.c of the reference) -- runs everywhere, no GPU, no reference tree."""
import json
import os

import numpy as np

import _libs

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))
PN = 0x83848D96BBCC54FC


def test_syndrome_vectors():
    orc = _libs.oracle()
    for cw, syn in VEC["syndrome"]["vectors"]:
        assert orc.orc_gen_syndrome(int(cw, 16)) == int(syn, 16)


def test_syncword_correction_vectors():
    """decode_syncword() of the stale test = syndrome lookup + xor error, then ^pn."""
    import ctypes as C
    orc = _libs.oracle()
    orc.orc_init(2)
    for cw, want in VEC["syncword_correct"]["vectors"]:
        cw = int(cw, 16)
        syn = orc.orc_gen_syndrome(cw)
        if syn:
            err = C.c_uint64()
            assert orc.orc_find_syndrome(syn, C.byref(err))
            cw ^= err.value
        assert cw ^ PN == int(want, 16)


def test_gen_syncword_vectors():
    orc = _libs.oracle()
    for lap, sw in VEC["gen_syncword"]["vectors"]:
        assert orc.orc_gen_syncword(int(lap, 16)) == int(sw, 16)


def test_syndrome_map_size():
    orc = _libs.oracle()
    orc.orc_init(2)
    assert orc.orc_syndrome_count() == 1711     # HASH_COUNT after btbb_init(2)


def test_fec23_vectors():
    orc = _libs.oracle()
    par = VEC["fec23_parity"]["vectors"]
    for i in range(10):
        clean = np.zeros(15, np.uint8)
        clean[i] = 1
        clean[10:] = par[i]
        want = np.zeros(10, np.uint8)
        want[i] = 1
        out = np.zeros(10, np.uint8)
        assert orc.orc_unfec23(_libs.ptr(clean), 1, _libs.ptr(out)) == 1
        assert (out == want).all()
        erased = clean.copy()
        erased[i] = 0
        out[:] = 0
        assert orc.orc_unfec23(_libs.ptr(erased), 1, _libs.ptr(out)) == 1
        assert (out == want).all()
        # encoder side
        assert orc.orc_fec23(1 << i) >> 10 == sum(b << j for j, b in enumerate(par[i]))


def test_hec_vectors():
    orc = _libs.oracle()
    for (uap, data, hec), octal in zip(VEC["hec"]["vectors"], VEC["header_fec13_octal"]["vectors"]):
        uap, data, hec = int(uap, 16), int(data, 16), int(hec, 16)
        assert orc.orc_uap_from_hec(data, hec) == uap
        assert orc.orc_hec_from_uap(data, uap) == hec
        # the octal column is the 18 header bits, LSB first, one digit per bit
        bits = [1 if c == "7" else 0 for c in octal]
        assert sum(b << i for i, b in enumerate(bits[:10])) == data
        assert sum(b << i for i, b in enumerate(bits[10:])) == hec
        # FEC 1/3 of those bits decodes back
        sym = np.repeat(np.array(bits, np.uint8), 3)
        out = np.zeros(18, np.uint8)
        assert orc.orc_unfec13(_libs.ptr(sym), _libs.ptr(out), 18) == 1
        assert out.tolist() == bits
