"""A numpy model of ONE WAVE decoding ONE long packet (NOTEBOOK.md 9, item 4) -- the arithmetic a wave-per-packet
`decode_hits` path for DH3 / DH5 / DM3 / DM5 would run, so that it can be checked against the oracle before a kernel
exists.  64 lanes x one 64-bit word is a whole 3 125-symbol capture:

  * DH (bluetooth_packet.c:962-1011): lane l owns payload bits [64 l, 64 l + 64): symbols 122 + 64 l .. XOR the
    whitening sequence from index start(clock) + 18 + 64 l (mod 127) -- no lane depends on another.
  * DM (:898-958): lane l owns FEC 2/3 blocks 3 l .. 3 l + 2 (45 symbols -> 30 data bits at payload bit 30 l); a
    payload word is then put together from the (at most four) 30-bit pieces that overlap it; whitening as above.
  * CRC (:772-781, register over ALL payload_length bytes == 0 <=> the reference's compare): the register is GF(2)-
    linear, and leading zero bytes do not move a zero register.  The payload is cut into 8-byte blocks aligned to its
    END (block l = the bytes that have exactly 8 l bytes behind them; the first block may be short), every lane runs
    its block from a zero register, advances the result over the 8 l bytes behind it -- a FIXED 16 x 16 bit matrix per
    lane, A^(64 l) -- and the wave XORs the 64 results with the seed advanced over the whole length.

Everything here is plain per-lane arithmetic plus two wave-wide operations (a 4-way gather of neighbours' pieces for
DM, an XOR reduction for the CRC).  tests/test_wave_decode_model.py compares it with the oracle."""
import numpy as np

from libbtbb_amd import synth

LANES = 64
_CAP = {3: 20, 4: 30, 8: 12, 10: 125, 11: 187, 14: 228, 15: 343}


def _crc_step_bits(reg, bits):
    for b in bits:
        fb = (reg & 1) ^ int(b)
        reg = (reg >> 1) | (fb << 15)
        reg ^= (reg & 0x8000) >> 5
        reg ^= (reg & 0x8000) >> 12
    return reg


def _advance_matrix(nbits):
    """Columns of the map reg -> register after nbits zero bits."""
    return [_crc_step_bits(1 << j, np.zeros(nbits, np.uint8)) for j in range(16)]


_ADV64 = None


def _adv64():
    """Lane l's constant: A^(64 l) as sixteen 16-bit columns (2 KiB for the wave)."""
    global _ADV64
    if _ADV64 is None:
        step = _advance_matrix(64)
        mats = [[1 << j for j in range(16)]]
        for _ in range(1, LANES):
            prev = mats[-1]
            mats.append([_apply(step, c) for c in prev])
        _ADV64 = mats
    return _ADV64


def _apply(cols, reg):
    out = 0
    for j in range(16):
        if (reg >> j) & 1:
            out ^= cols[j]
    return out


def _int_of(bits):
    return sum(int(b) << j for j, b in enumerate(bits))


def _seed(uap):
    return (int("{:08b}".format(uap & 0xFF)[::-1], 2) << 8) & 0xFF00


def _payload_header(raw_bits, clk6, ptype, two_bytes):
    hbits = 16 if two_bytes else 8
    ph = raw_bits[:hbits] ^ synth.whitening(clk6, 18, hbits)
    field = sum(int(b) << k for k, b in enumerate(ph[3:13 if two_bytes else 8]))
    plen = field + (4 if two_bytes else 3)
    return min(plen, _CAP.get(ptype, 0))


def _wave_crc_is_zero(out_words, nbits, uap):
    """out_words[l] = payload bits 64 l .. as a uint8 bit array of 64; nbits a multiple of 8."""
    allbits = np.concatenate(out_words)[:nbits]
    adv = _adv64()
    total = _crc_step_bits(_seed(uap), np.zeros(nbits, np.uint8))      # the seed over the whole length (one table row)
    nblocks = (nbits + 63) // 64
    for lane in range(nblocks):                                        # block `lane` has 64 * lane bits behind it
        hi = nbits - 64 * lane
        lo = max(0, hi - 64)
        # what the lane holds: a funnel shift of two neighbouring output words by nbits % 64
        reg = _crc_step_bits(0, allbits[lo:hi])
        total ^= _apply(adv[lane], reg)                                # per-lane constant matrix, then the wave XOR
    return total == 0


def dh_wave(sym, clk6, uap, ptype):
    """(rv, payload bits or None) of orc_DH / do_DH for one packet; sym = captured symbols (uint8 0/1)."""
    two = ptype in (11, 15)
    max_length = {4: 30, 9: 30, 11: 187, 15: 343}[ptype]
    size = len(sym) - 122
    hbits = 16 if two else 8
    if size < hbits:
        return 0, None
    s = np.concatenate([sym, np.zeros(64 * LANES + 128, np.uint8)])
    plen = _payload_header(s[122:122 + hbits], clk6, ptype, two)
    if plen > max_length:
        return 1, None
    nbits = plen * 8
    if nbits > size:
        return 1, None
    words = []
    for lane in range((nbits + 63) // 64):
        raw = s[122 + 64 * lane:122 + 64 * lane + 64]
        w = raw ^ synth.whitening(clk6, 18 + 64 * lane, 64)
        keep = min(64, nbits - 64 * lane)
        w[keep:] = 0
        words.append(w)
    payload = np.concatenate(words)[:nbits] if words else np.zeros(0, np.uint8)
    if ptype == 9:
        return 2, payload
    ok = _wave_crc_is_zero(words, nbits, uap)
    assert ok == wave_crc_is_zero_u64([_int_of(w) for w in words], nbits, uap)
    assert ok == wave_crc_is_zero_start_aligned([_int_of(w) for w in words], nbits, uap)
    assert ok == wave_crc_is_zero_two_words_per_lane([_int_of(w) for w in words], nbits, uap)
    return (10 if ok else 2), payload


_F23 = synth._F23


def _fec23_block(blk15):
    """15 symbols -> (ok, 10 corrected data bits): single-error correction of the (15,10) code (:602-646)."""
    data = blk15[:10].copy()
    par = 0
    for i in range(10):
        if data[i]:
            par ^= _F23[i]
    syn = par ^ sum(int(b) << k for k, b in enumerate(blk15[10:15]))
    if syn & (syn - 1) == 0:                     # nothing wrong, or one of the five check symbols
        return True, data
    for i in range(10):
        if syn == _F23[i]:
            data[i] ^= 1
            return True, data
    return False, data


def dm_wave(sym, clk6, uap, ptype):
    """(rv, payload bits or None) of orc_DM / do_DM for one packet."""
    two = ptype in (10, 14)
    max_length = {3: 20, 8: 12, 10: 125, 14: 228}[ptype]
    pos = 202 if ptype == 8 else 122
    size = len(sym) - pos
    hbits = 16 if two else 8
    if size < hbits or size < (30 if two else 15):
        return 0, None
    s = np.concatenate([sym, np.zeros(64 * LANES + 256, np.uint8)])
    hdr = []
    for b in range(2 if two else 1):
        ok, d = _fec23_block(s[pos + 15 * b:pos + 15 * b + 15])
        if not ok:
            return 0, None
        hdr.append(d)
    plen = _payload_header(np.concatenate(hdr), clk6, ptype, two)
    if plen > max_length:
        return 1, None
    nbits = plen * 8
    if nbits > size:
        return 1, None
    nblocks = (nbits + 9) // 10
    # lane l: blocks 3 l .. 3 l + 2 -> a 30-bit piece (a wave-wide "any block failed" decides rv 0)
    pieces, failed = [], False
    for lane in range((nblocks + 2) // 3):
        piece = np.zeros(30, np.uint8)
        for j in range(3):
            b = 3 * lane + j
            if b >= nblocks:
                break
            ok, d = _fec23_block(s[pos + 15 * b:pos + 15 * b + 15])
            failed |= not ok
            piece[10 * j:10 * j + 10] = d
        pieces.append(piece)
    if failed:
        return 0, None
    words = []
    for lane in range((nbits + 63) // 64):                             # word l from the pieces that overlap it
        w = np.zeros(64, np.uint8)
        first, last = (64 * lane) // 30, (64 * lane + 63) // 30
        assert last - first <= 3
        for p in range(first, min(last, len(pieces) - 1) + 1):
            for k in range(30):
                bit = 30 * p + k - 64 * lane
                if 0 <= bit < 64:
                    w[bit] = pieces[p][k]
        w ^= synth.whitening(clk6, 18 + 64 * lane, 64)
        keep = min(64, nbits - 64 * lane)
        w[keep:] = 0
        words.append(w)
    payload = np.concatenate(words)[:nbits] if words else np.zeros(0, np.uint8)
    ok = _wave_crc_is_zero(words, nbits, uap)
    assert ok == wave_crc_is_zero_u64([_int_of(w) for w in words], nbits, uap)
    assert ok == wave_crc_is_zero_start_aligned([_int_of(w) for w in words], nbits, uap)
    assert ok == wave_crc_is_zero_two_words_per_lane([_int_of(w) for w in words], nbits, uap)
    return (10 if ok else 2), payload


def _bits_of(word):
    return np.array([(word >> j) & 1 for j in range(64)], dtype=np.uint8)


def wave_crc_is_zero_u64(words, nbits, uap):
    """The same with the registers a lane would hold: words[w] = payload bits 64 w .. 64 w + 63 as an integer (bit j =
    payload bit 64 w + j, zero behind nbits).  With nbits = 64 T + r, lane l's block is the funnel shift
    (words[T - l - 1] >> r) | (words[T - l] << (64 - r)) of two neighbouring words (words[-1] = 0; r = 0: words[T - 1 - l]),
    i.e. the 64 payload bits that end 64 l bits in front of the end; the first block starts with zeros."""
    mask = (1 << 64) - 1
    T, r = divmod(nbits, 64)
    get = lambda i: words[i] if 0 <= i < len(words) else 0          # noqa: E731
    adv = _adv64()
    total = _crc_step_bits(_seed(uap), np.zeros(nbits, np.uint8))
    for lane in range((nbits + 63) // 64):
        if r:
            block = ((get(T - lane - 1) >> r) | (get(T - lane) << (64 - r))) & mask
        else:
            block = get(T - 1 - lane)
        total ^= _apply(adv[lane], _crc_step_bits(0, _bits_of(block)))
    return total == 0


_ADV64INV = None


def _adv64inv():
    """Lane l's constant in the kernel as built (csrc/packet.hip long_payloads): A^(-64 l) as sixteen 16-bit columns, A = one
    zero bit through the register.  Found like the library finds it: run every register value forward over 64 zero bits and
    read the map backwards."""
    global _ADV64INV
    if _ADV64INV is None:
        step = _advance_matrix(64)
        back = {}
        for u in range(1 << 16):
            back[_apply(step, u)] = u
        assert len(back) == 1 << 16                                   # the step is invertible
        inv = [back[1 << j] for j in range(16)]
        mats = [[1 << j for j in range(16)]]
        for _ in range(1, LANES):
            mats.append([_apply(inv, c) for c in mats[-1]])
        _ADV64INV = mats
    return _ADV64INV


def wave_crc_is_zero_start_aligned(words, nbits, uap):
    """The form the kernel ran until round 4 (one word per lane): no lane needs another lane's word.  words[w] as in wave_crc_is_zero_u64 (zero behind nbits).
    Appending zero bits advances the register by an invertible map, so `register == 0` may be tested on the payload padded
    to whole words; the seed is its bits on the first sixteen message bits; and the register after n words is A^(64 (n - 1))
    of the XOR over the words of A^(-64 w) (register of word w alone) -- again an invertible outer factor."""
    inv = _adv64inv()
    total = 0
    for lane, word in enumerate(words):
        if lane == 0:
            word ^= _seed(uap)
        total ^= _apply(inv[lane], _crc_step_bits(0, _bits_of(word)))
    return total == 0


def wave_crc_is_zero_two_words_per_lane(words, nbits, uap):
    """The kernel as of round 4 (long_payloads): lane l runs payload words 2 l and 2 l + 1 through ONE register from zero
    (128 steps; a missing second word is 64 zero bits) and applies A^(-128 l), row 2 l of the same table: the XOR over the
    lanes is A^64 of wave_crc_is_zero_start_aligned's total -- zero exactly when that is."""
    inv = _adv64inv()
    total = 0
    words = list(words) + [0] * (len(words) & 1)
    for lane in range(len(words) // 2):
        w0 = words[2 * lane] ^ (_seed(uap) if lane == 0 else 0)
        reg = _crc_step_bits(_crc_step_bits(0, _bits_of(w0)), _bits_of(words[2 * lane + 1]))
        total ^= _apply(inv[2 * lane], reg)
    return total == 0


def wave_crc_is_zero_three_words_per_lane(words, nbits, uap):
    """dh_payloads: lane l runs payload words 3 l .. 3 l + 2 through one register from zero and applies A^(-192 l) (row 3 l)."""
    inv = _adv64inv()
    total = 0
    words = list(words) + [0] * (-len(words) % 3)
    for lane in range(len(words) // 3):
        reg = 0
        for k in range(3):
            w = words[3 * lane + k] ^ (_seed(uap) if lane == 0 and k == 0 else 0)
            reg = _crc_step_bits(reg, _bits_of(w))
        total ^= _apply(inv[3 * lane], reg)
    return total == 0


def ev_registers_by_lane_prefix(words, uap):
    """ev_payloads: the CRC register IN FRONT of every payload word without a lane walking the words in front of it.
    q_j = A^(-64 (j + 1)) (register of word j alone, the seed on word 0's first sixteen bits); the register in front of word l
    is A^(64 l) applied to the XOR of q_0 .. q_(l - 1) -- an exclusive XOR prefix over the lanes between two per-lane
    matrices (g_adv64inv row l + 1, g_adv64fwd row l).  -> list of registers, one per word"""
    inv, fwd = _adv64inv(), _adv64()
    out, acc = [], 0
    for lane, word in enumerate(words):
        out.append(_apply(fwd[lane], acc))
        w = word ^ (_seed(uap) if lane == 0 else 0)
        acc ^= _apply(inv[lane + 1], _crc_step_bits(0, _bits_of(w)))
    return out


def fec23_quad(stream_bits, q, have):
    """long_payloads' step 2b for ONE lane: the four (15,10) blocks that start at stream bit q, `have` of them inside the
    packet -> (40 payload bits as an integer, any undecodable).  Written the way the kernel indexes: three dwords, two
    funnel shifts, the third block across the 32-bit seam."""
    dw = [_int_of(stream_bits[32 * i:32 * i + 32]) for i in range((len(stream_bits) + 31) // 32)] + [0, 0, 0]
    i, s = q >> 5, q & 31
    align = lambda hi, lo, sh: ((lo >> sh) | (hi << (32 - sh))) & 0xFFFFFFFF if sh else lo      # noqa: E731
    vm = (1 << (15 * have)) - 1
    x0 = align(dw[i + 1], dw[i], s) & (vm & 0xFFFFFFFF)
    x1 = align(dw[i + 2], dw[i + 1], s) & (vm >> 32)
    b2 = align(x1, x0, 30)
    blocks = [x0 & 0x7FFF, (x0 >> 15) & 0x7FFF, b2 & 0x7FFF, (x1 >> 13) & 0x7FFF]
    out, bad = 0, False
    for k, blk in enumerate(blocks):
        bits15 = np.array([(blk >> j) & 1 for j in range(15)], dtype=np.uint8)
        ok, data = _fec23_block(bits15)
        if k < have and not ok:
            bad = True
        out |= _int_of(data) << (10 * k)
    return out, bad
