"""Helpers shared by packet-level parity tests: build the same packet in the oracle and in
the compiled reference, and compare every field the hot path touches."""
import ctypes as C

import numpy as np

import _libs
from libbtbb_amd import synth

FIELDS = [  # (name, dtype, count)
    ("flags", "<u4", 1), ("UAP", "u1", 1), ("LAP", "<u4", 1), ("packet_type", "u1", 1),
    ("packet_lt_addr", "u1", 1), ("packet_flags", "u1", 1), ("packet_hec", "u1", 1),
    ("packet_header", "u1", 18), ("payload_header_length", "<i4", 1), ("payload_header", "u1", 16),
    ("payload_llid", "u1", 1), ("payload_flow", "u1", 1), ("payload_length", "<i4", 1),
    ("payload", "u1", 2744), ("clkn", "<u4", 1), ("ac_errors", "u1", 1), ("length", "<u2", 1),
]


def orc_state(p):
    """dict of numpy values from an OrcPacket pointer."""
    s = p.contents
    out = {}
    for name, dt, cnt in FIELDS:
        v = getattr(s, name)
        if cnt > 1:
            out[name] = np.frombuffer(bytes(v) if not isinstance(v, bytes) else v.ljust(cnt, b"\0"), dtype="u1")[:cnt].copy()
            # c_char arrays are returned as bytes truncated at NUL: read raw memory instead
            off = getattr(type(s), name).offset
            out[name] = np.frombuffer((C.c_uint8 * cnt).from_address(C.addressof(s) + off), dtype="u1").copy()
        else:
            out[name] = int(v)
    return out


def ref_state(ref, p):
    view = _libs.RefPacketView(ref, p.value if isinstance(p, C.c_void_p) else p)
    out = {}
    for name, dt, cnt in FIELDS:
        v = view.field(name, dt, cnt)
        out[name] = v.copy() if cnt > 1 else int(v)
    return out


def assert_same(a, b, ctx=""):
    for name, _, cnt in FIELDS:
        if cnt > 1:
            assert (a[name] == b[name]).all(), (ctx, name, np.nonzero(a[name] != b[name])[0][:8])
        else:
            assert a[name] == b[name], (ctx, name, a[name], b[name])


class Pair:
    """The same packet object in the oracle and in the reference."""

    def __init__(self, orc, ref, lap=0, ac_errors=0):
        self.orc, self.ref = orc, ref
        self.o = orc.orc_packet_new()
        self.r = C.c_void_p(ref.btbb_packet_new())
        orc.orc_packet_init_found(self.o, lap, ac_errors)
        # reference: init_packet is static; same effect through the public setters
        view = _libs.RefPacketView(ref, self.r.value)
        C.c_uint32.from_address(self.r.value + view._off("LAP")).value = lap
        C.c_uint8.from_address(self.r.value + view._off("ac_errors")).value = ac_errors
        C.c_uint32.from_address(self.r.value + view._off("flags")).value = 0
        ref.btbb_packet_set_flag(self.r, 0, 1)

    def set_data(self, sym, channel=0, clkn=0):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        self.orc.orc_packet_set_data(self.o, _libs.ptr(sym), len(sym), channel, clkn)
        self.ref.btbb_packet_set_data(self.r, _libs.ptr(sym), len(sym), channel, clkn)

    def set_flag(self, flag, val):
        self.orc.orc_packet_set_flag(self.o, flag, val)
        self.ref.btbb_packet_set_flag(self.r, flag, val)

    def set_uap(self, uap):
        self.o.contents.UAP = uap
        self.orc.orc_packet_set_flag(self.o, 2, 1)
        self.ref.btbb_packet_set_uap(self.r, uap)

    def check(self, ctx=""):
        assert_same(orc_state(self.o), ref_state(self.ref, self.r), ctx)

    def close(self):
        self.orc.orc_packet_free(self.o)
        self.ref.btbb_packet_unref(self.r)


def random_packets(rng, n, max_sym_errors=3):
    """A mix of well-formed packets of every type (with a few symbol errors) and junk with
    a valid FEC-1/3 header, as (symbols, meta) tuples."""
    out = []
    types = list(range(16))
    for i in range(n):
        lap = int(rng.integers(0, 1 << 24))
        uap = int(rng.integers(0, 256))
        clk6 = int(rng.integers(0, 64))
        kind = i % 3
        if kind < 2:
            t = types[(i // 3) % 16] if kind == 0 else int(rng.integers(0, 16))
            maxlen = {3: 17, 4: 27, 8: 9, 9: 29, 10: 121, 11: 183, 14: 224, 15: 339, 5: 10, 6: 20, 7: 30,
                      12: 120, 13: 180}.get(t, 0)
            if t in (5, 6, 7):
                body = rng.integers(0, 256, maxlen, dtype=np.uint8).tobytes()
            else:
                body = rng.integers(0, 256, int(rng.integers(0, maxlen + 1)), dtype=np.uint8).tobytes()
            sym = synth.build_packet(lap, uap, clk6, t, lt_addr=int(rng.integers(0, 8)), flags=int(rng.integers(0, 8)),
                                     body=body, llid=int(rng.integers(0, 4)), flow=int(rng.integers(0, 2)),
                                     voice=rng.integers(0, 256, 10, dtype=np.uint8).tobytes(),
                                     fhs_bits=synth.fhs_payload(lap, uap, int(rng.integers(0, 1 << 16)),
                                                                int(rng.integers(0, 1 << 26)), rng))
            tail = rng.integers(0, 2, int(rng.integers(0, 400)), dtype=np.uint8)
            sym = np.concatenate([sym, tail])
            ne = int(rng.integers(0, max_sym_errors + 1))
            if ne:
                sym[rng.integers(64, len(sym), ne)] ^= 1
            if rng.random() < 0.15:                 # truncated capture
                sym = sym[: int(rng.integers(60, len(sym) + 1))]
        else:
            L = int(rng.integers(122, 3400))
            sym = rng.integers(0, 2, L, dtype=np.uint8)
            sym[:68] = synth.access_code(lap)
            hdr = rng.integers(0, 2, 18, dtype=np.uint8)
            sym[68:122] = np.repeat(hdr, 3)
            if rng.random() < 0.3:
                sym[rng.integers(68, 122, int(rng.integers(1, 6)))] ^= 1
            t = -1
        out.append((np.ascontiguousarray(sym[:4000]), dict(lap=lap, uap=uap, clk6=clk6, type=t)))
    return out
