"""Shared scenario builders for the hop-sequence / CLK1-27 reversal tests (test infrastructure)."""
import ctypes as C

import numpy as np

import _libs

SEQ_LEN = 1 << 27
F_UAP_VALID, F_CLK6_VALID, F_CLK27_VALID, F_HOP_INIT, F_GOT_FIRST, F_IS_AFH, F_LOOKS_AFH, F_ALIASED, F_FOLLOWING = \
    2, 4, 5, 9, 10, 11, 12, 13, 14


def afh_map_bytes(rng, n_used):
    """10-byte AFH map with n_used of the 79 channels set."""
    chans = rng.choice(79, size=n_used, replace=False)
    m = np.zeros(10, np.uint8)
    for c in chans:
        m[c // 8] |= 1 << (c % 8)
    return m


def orc_pattern(orc, lap, uap, afh_map=None):
    """A fresh oracle piconet with its hop pattern fetched; returns (pn, sequence view)."""
    pn = orc.orc_piconet_new()
    orc.orc_init_piconet(pn, lap)
    pn.contents.UAP = uap
    orc.orc_piconet_set_flag(pn, F_UAP_VALID, 1)
    if afh_map is not None:
        orc.orc_piconet_set_flag(pn, F_IS_AFH, 1)
        orc.orc_piconet_set_afh_map(pn, _libs.ptr(afh_map))    # fetches the pattern (UAP valid)
    else:
        # H6: the reference's gen_hops divides by used_channels even without AFH, so a piconet
        # that has seen no channel yet crashes there; real callers have always seen one
        pn.contents.afh_map[0] |= 1
        pn.contents.used_channels = 1
        orc.orc_get_hop_pattern(pn)
    return pn, seq_view(pn.contents.sequence)


def seq_view(address):
    """numpy view of a 2^27-byte sequence living at a C address."""
    buf = (C.c_uint8 * SEQ_LEN).from_address(address)
    return np.frombuffer(buf, dtype=np.uint8)


def aliased(ch):
    return ((int(ch) + 24) % 25) + 26


def observations(rng, seq, c0, n, alias=False, max_gap=400):
    """n observed hops of a piconet whose CLK1-27 was c0 at the first one: [(index, channel)]."""
    out, d = [], 0
    for k in range(n):
        ch = int(seq[(c0 + d) % SEQ_LEN])
        out.append((d, aliased(ch) if alias else ch))
        d += int(rng.integers(1, max_gap))
    return out


def piconet_traffic(rng, seq, lap, uap, c0, n, local_offset=34, max_gap=300, types=(0, 1, 9, 3, 4, 10)):
    """n packets of one piconet as a receiver would hand them to btbb_process_packet:
    [(symbols, channel, clkn)].  c0 = master CLK1-27 at the first packet; the receiver's own
    clock runs `local_offset` half-slots ahead of the master's 28-bit clock."""
    from libbtbb_amd import synth
    out, c = [], c0
    for k in range(n):
        t = int(types[int(rng.integers(0, len(types)))])
        body = rng.integers(0, 256, 9, dtype=np.uint8).tobytes()
        sym = synth.build_packet(lap, uap, c & 0x3F, t, lt_addr=1, body=body,
                                 fhs_bits=synth.fhs_payload(lap, uap, 1, 2, rng))
        sym = np.concatenate([sym, rng.integers(0, 2, 40, dtype=np.uint8)])
        clkn = ((c << 1) + local_offset) & 0xFFFFFFF
        out.append((np.ascontiguousarray(sym), int(seq[c % SEQ_LEN]), clkn))
        c = (c + int(rng.integers(1, max_gap))) % SEQ_LEN
    return out
