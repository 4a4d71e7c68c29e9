"""Synthetic Bluetooth BR baseband traffic (host side, numpy).

The reference library contains no transmitter (tests/test_header.c:47 refers to a
``bluetooth_packet_tx.h`` that does not exist), so everything the parity tests and the
benchmark feed into the scan/decode path is built here from the Bluetooth baseband
spec: sync-word (64,30) block code, whitening LFSR, FEC 1/3, FEC 2/3, HEC and CRC-16.
SURVEY.md Appendix B lists the air format and the reference lines that consume it.

Also here: the counter-based PRNG that defines the synthetic bit streams of
BASELINE.json's configs.  The HIP generator (csrc/synth.hip) implements the same
functions, so any slice of a device-generated stream can be regenerated on the host.

Nothing in this module touches the GPU or the oracle.
"""
import numpy as np

# ----------------------------------------------------------------------------------
# spec constants
# ----------------------------------------------------------------------------------
SW_POLY = 0o260534236651           # (64,30) code generator, degree 34
SW_PN = 0x83848D96BBCC54FC         # PN overlay
LAP_ANY = 0xFFFFFFFF
MAX_SYMBOLS = 3125

TYPE_NULL, TYPE_POLL, TYPE_FHS, TYPE_DM1, TYPE_DH1, TYPE_HV1, TYPE_HV2, TYPE_HV3 = range(8)
TYPE_DV, TYPE_AUX1, TYPE_DM3, TYPE_DH3, TYPE_EV4, TYPE_EV5, TYPE_DM5, TYPE_DH5 = range(8, 16)

_M64 = (1 << 64) - 1


def _sw_cols():
    cols, c = [], 1
    for _ in range(64):
        cols.append(c)
        c <<= 1
        if c >> 34 & 1:
            c ^= SW_POLY
    return cols


_COLS = _sw_cols()


def syncword(lap):
    """64-bit sync word (host order: bit i = i-th transmitted symbol) of a 24-bit LAP."""
    lap &= 0xFFFFFF
    info = ((0x13 if lap & 0x800000 else 0x2C) << 24) | lap
    info ^= SW_PN >> 34
    cw = 0
    for b in range(30):
        if info >> b & 1:
            cw ^= (1 << (34 + b)) | _COLS[34 + b]
    return cw ^ SW_PN


def _whitening_period():
    seq, states, s = [], {}, 0x7F
    for i in range(127):
        out = s >> 6 & 1
        states[s] = i
        seq.append(out)
        s = (s << 1) & 0x7F
        if out:
            s ^= 0x11
    return np.array(seq, dtype=np.uint8), states


_WSEQ, _WSTATE = _whitening_period()


def whitening(clk6, skip, n):
    """n whitening bits for CLK1-6 = clk6, starting `skip` bits into the packet (header = 0)."""
    start = _WSTATE[0x40 | (clk6 & 0x3F)] + skip
    return _WSEQ[(start + np.arange(n)) % 127]


def bits_lsb(value, n):
    return np.array([(value >> i) & 1 for i in range(n)], dtype=np.uint8)


def bytes_to_bits(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    return np.unpackbits(a, bitorder="little")


def _rev8(b):
    return int("{:08b}".format(b & 0xFF)[::-1], 2)


def hec(data10, uap):
    """HEC for the 10 header bits (LT_ADDR|TYPE|FLOW|ARQN|SEQN, LSB first)."""
    reg = _rev8(uap)
    for i in range(10):
        msb = (reg & 1) ^ (data10 >> i & 1)
        reg = (reg >> 1) | (msb << 7)
        if msb:
            reg ^= 0x65
    return reg


def crc16(bits, uap):
    """CRC-CCITT over an air-order bit array, register seeded with the UAP."""
    reg = (_rev8(uap) << 8) & 0xFF00
    for b in bits:
        fb = (reg & 1) ^ int(b)
        reg = (reg >> 1) | (fb << 15)
        reg ^= (reg & 0x8000) >> 5
        reg ^= (reg & 0x8000) >> 12
    return reg


def fec13(bits):
    return np.repeat(np.asarray(bits, dtype=np.uint8), 3)


def _fec23_parity_cols():
    cols = []
    for i in range(10):
        p = 1 << (14 - i)
        for k in range(14, 4, -1):
            if p >> k & 1:
                p ^= 0x35 << (k - 5)
        cols.append(int("{:05b}".format(p)[::-1], 2))
    return cols


_F23 = _fec23_parity_cols()


def fec23(bits):
    """(15,10) shortened Hamming code, zero padded to a multiple of 10 data bits."""
    bits = np.asarray(bits, dtype=np.uint8)
    pad = (-len(bits)) % 10
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, np.uint8)])
    out = []
    for k in range(0, len(bits), 10):
        d = bits[k:k + 10]
        par = 0
        for i in range(10):
            if d[i]:
                par ^= _F23[i]
        out.append(d)
        out.append(bits_lsb(par, 5))
    return np.concatenate(out)


# ----------------------------------------------------------------------------------
# packets
# ----------------------------------------------------------------------------------
_FEC23_TYPES = {TYPE_FHS, TYPE_DM1, TYPE_DM3, TYPE_DM5, TYPE_DV, TYPE_HV2, TYPE_EV4}
_TWO_BYTE_HDR = {TYPE_DM3, TYPE_DH3, TYPE_DM5, TYPE_DH5}
_ACL_CRC = {TYPE_DM1, TYPE_DH1, TYPE_DM3, TYPE_DH3, TYPE_DM5, TYPE_DH5, TYPE_DV}


def access_code(lap, trailer=True):
    sw = syncword(lap)
    bits = bits_lsb(sw, 64)
    if trailer:
        m = bits[63]
        bits = np.concatenate([bits, np.array([1 - m, m, 1 - m, m], np.uint8)])
    return bits


def header_bits(lt_addr, ptype, flags, uap, clk6):
    data = (lt_addr & 7) | (ptype & 0xF) << 3 | (flags & 7) << 7
    h = np.concatenate([bits_lsb(data, 10), bits_lsb(hec(data, uap), 8)])
    return fec13(h ^ whitening(clk6, 0, 18))


def fhs_payload(lap, uap, nap, clk27_2, rng=None):
    """18 FHS data bytes as a 144-bit array with the address fields the reference extracts
    (bluetooth_packet.c:1411-1441): LAP at 34..57, UAP 64..71, NAP 72..87, CLK 115..140."""
    rng = rng or np.random.default_rng(0)
    bits = rng.integers(0, 2, 144, dtype=np.uint8)
    bits[34:58] = bits_lsb(lap, 24)
    bits[64:72] = bits_lsb(uap, 8)
    bits[72:88] = bits_lsb(nap, 16)
    bits[115:141] = bits_lsb(clk27_2, 26)
    return bits


def build_packet(lap, uap=0, clk6=0, ptype=None, lt_addr=1, flags=0, body=b"", llid=2, flow=1,
                 voice=None, fhs_bits=None):
    """Air-order symbols (one 0/1 byte each), index 0 = first sync-word bit.

    ptype None -> ID packet (64 symbols).  `body` = user payload bytes (no payload header,
    no CRC).  For HV1/2/3 `body` is the 10/20/30 voice bytes; for DV `voice` is the 10 voice
    bytes and `body` the data field; for EV3/4/5 body is payload, CRC appended.
    """
    if ptype is None:
        return access_code(lap, trailer=False)
    parts = [access_code(lap), header_bits(lt_addr, ptype, flags, uap, clk6)]
    skip = 18
    pl = None
    if ptype in (TYPE_NULL, TYPE_POLL):
        pass
    elif ptype == TYPE_FHS:
        data = fhs_bits if fhs_bits is not None else fhs_payload(lap, uap, 0, 0)
        assert len(data) == 144
        pl = np.concatenate([data, bits_lsb(crc16(data, uap), 16)])
    elif ptype in _ACL_CRC or ptype == TYPE_AUX1:
        n = len(body)
        if ptype in _TWO_BYTE_HDR:
            hdr = bits_lsb((llid & 3) | (flow & 1) << 2 | (n & 0x3FF) << 3, 16)
        else:
            hdr = bits_lsb((llid & 3) | (flow & 1) << 2 | (n & 0x1F) << 3, 8)
        data = np.concatenate([hdr, bytes_to_bits(body)])
        if ptype != TYPE_AUX1:
            data = np.concatenate([data, bits_lsb(crc16(data, uap), 16)])
        pl = data
    elif ptype in (TYPE_HV1, TYPE_HV2, TYPE_HV3):
        pl = bytes_to_bits(body)
    elif ptype in (TYPE_EV4, TYPE_EV5):
        data = bytes_to_bits(body)
        pl = np.concatenate([data, bits_lsb(crc16(data, uap), 16)])
    else:
        raise ValueError(ptype)

    if ptype == TYPE_DV:
        v = bytes_to_bits(voice if voice is not None else bytes(10))
        assert len(v) == 80
        # the reference decodes the data field with whitening index 18 straight after the
        # header (bluetooth_packet.c:913-916, 937): the voice field is whitened separately
        parts.append(v ^ whitening(clk6, skip, 80))
    if pl is not None:
        w = pl ^ whitening(clk6, skip, len(pl))
        if ptype in _FEC23_TYPES:
            w = fec23(w)
        elif ptype == TYPE_HV1:
            w = fec13(w)
        parts.append(w)
    return np.concatenate(parts).astype(np.uint8)


# ----------------------------------------------------------------------------------
# packed bit streams and the counter-based generator
# ----------------------------------------------------------------------------------
def pack_bits(sym):
    """0/1 bytes -> LSB-first uint64 words (stream bit i = bit i%64 of word i//64)."""
    sym = np.asarray(sym, dtype=np.uint8)
    pad = (-len(sym)) % 64
    if pad:
        sym = np.concatenate([sym, np.zeros(pad, np.uint8)])
    return np.packbits(sym, bitorder="little").view("<u8").copy()


def unpack_bits(words, nbits=None):
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
    return b if nbits is None else b[:nbits]


def splitmix64(x):
    """Vectorised splitmix64 of uint64 counters."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def noise_words(seed, first_word, nwords):
    """iid Bernoulli(1/2) stream words: word[i] = splitmix64(seed + i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(first_word, first_word + nwords, dtype=np.uint64) + np.uint64(seed & _M64)
    return splitmix64(idx)


INJECT_SALT = 0xA5A5A5A55A5A5A5A


def injection_params(seed, k, stride, err_cycle=4):
    """Position / LAP / error mask of injections k (array) for a stream with one ID sync word
    every `stride` symbols.  Errors (k % err_cycle of them) fall in sync-word bits 0..56."""
    k = np.asarray(k, dtype=np.uint64)
    assert stride >= 512
    with np.errstate(over="ignore"):
        base = np.uint64((seed ^ INJECT_SALT) & _M64) + k * np.uint64(4)
        h0 = splitmix64(base)
        h1 = splitmix64(base + np.uint64(1))
        h2 = splitmix64(base + np.uint64(2))
        pos = k * np.uint64(stride) + np.uint64(64) + h0 % np.uint64(stride - 256)
        lap = (h1 & np.uint64(0xFFFFFF)).astype(np.uint32)
        nerr = (k % np.uint64(err_cycle)).astype(np.int64)
        mask = np.zeros(k.shape, dtype=np.uint64)
        for j in range(5):
            e = (h2 >> np.uint64(8 * j)) & np.uint64(0xFF)
            e = e % np.uint64(57)
            mask ^= np.where(nerr > j, np.uint64(1) << e, np.uint64(0))
    return pos, lap, nerr, mask


def make_stream(seed, nwords, stride=4096, first_word=0, lap=None, err_cycle=4):
    """Packed synthetic stream [first_word, first_word+nwords) of the global stream `seed`:
    noise plus one sync word per `stride` symbols (random LAPs, or `lap` for all).
    Returns (words, injections) with injections = (pos, lap, nerr, mask) of those that start
    inside the slice (a sync word cut by the slice end is still written as far as it fits)."""
    words = noise_words(seed, first_word, nwords)
    lo, hi = first_word * 64, (first_word + nwords) * 64
    k0 = max(lo // stride - 1, 0)
    k1 = hi // stride + 1
    k = np.arange(k0, k1, dtype=np.uint64)
    pos, laps, nerr, mask = injection_params(seed, k, stride, err_cycle)
    if lap is not None:
        laps = np.full(laps.shape, lap, dtype=np.uint32)
    keep = []
    for i in range(len(k)):
        p = int(pos[i])
        if p + 64 <= lo or p >= hi:
            continue
        sw = syncword(int(laps[i])) ^ int(mask[i])
        for half in range(2):
            widx = p // 64 + half - first_word
            if widx < 0 or widx >= nwords:
                continue
            sh = p % 64
            if half == 0:
                m = (_M64 << sh) & _M64
                v = (sw << sh) & _M64
            else:
                if sh == 0:
                    continue
                m = _M64 >> (64 - sh)
                v = sw >> (64 - sh)
            words[widx] = np.uint64((int(words[widx]) & ~m & _M64) | v)
        if lo <= p < hi:
            keep.append(i)
    keep = np.array(keep, dtype=np.int64)
    return words, (pos[keep], laps[keep], nerr[keep], mask[keep])
