"""ctypes loaders for the parity checkers used by the test-suite.

* ``oracle()``  -> oracle/liboracle.so   (our CPU restatement; built on demand with gcc)
* ``ref()``     -> oracle/_ref/libbtbb_ref.so (the unmodified reference, compiled by
  oracle/Makefile from /root/reference when that tree is present; on the GPU box the
  prebuilt file travels with the snapshot).  Returns None if unavailable.

Test infrastructure only: nothing in libbtbb_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import functools

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

LAP_ANY = 0xFFFFFFFF
MAX_SYMBOLS = 3125


class OrcHit(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("lap", C.c_uint32), ("ac_errors", C.c_uint8), ("pad", C.c_uint8 * 3)]


class OrcPacket(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("channel", C.c_uint8), ("UAP", C.c_uint8), ("NAP", C.c_uint16),
        ("LAP", C.c_uint32), ("packet_type", C.c_uint8), ("packet_lt_addr", C.c_uint8),
        ("packet_flags", C.c_uint8), ("packet_hec", C.c_uint8), ("packet_header", C.c_char * 18),
        ("payload_header_length", C.c_int), ("payload_header", C.c_char * 16),
        ("payload_llid", C.c_uint8), ("payload_flow", C.c_uint8), ("payload_length", C.c_int),
        ("payload", C.c_char * 2744), ("clkn", C.c_uint32), ("ac_errors", C.c_uint8),
        ("length", C.c_uint16), ("symbols", C.c_char * 3125),
    ]


class OrcPiconet(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("afh_map", C.c_uint8 * 10), ("used_channels", C.c_uint8),
        ("LAP", C.c_uint32), ("UAP", C.c_uint8), ("packets_observed", C.c_int),
        ("total_packets_observed", C.c_int), ("clock6_candidates", C.c_int * 64),
        ("pattern_indices", C.c_int * 1000), ("pattern_channels", C.c_uint8 * 1000),
        ("clk_offset", C.c_int), ("first_pkt_time", C.c_uint32), ("hop_reversal_requests", C.c_int),
        ("aliased", C.c_int), ("a1", C.c_int), ("b", C.c_int), ("c1", C.c_int), ("d1", C.c_int), ("e", C.c_int),
        ("bank", C.c_int * 79), ("sequence", C.c_void_p), ("clock_candidates", C.POINTER(C.c_uint32)),
        ("num_candidates", C.c_int), ("winnowed", C.c_int),
    ]


def seed(n):
    """Test seeds are fixed; BTBB_TEST_SEED=k shifts all of them (soak runs on the GPU box)."""
    return int(n) + 100003 * int(os.environ.get("BTBB_TEST_SEED", "0"))


def build_oracle():
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)


@functools.lru_cache(maxsize=None)
def oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("btbb_oracle.c", "btbb_oracle_hop.c", "btbb_oracle.h")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
        build_oracle()
    lib = C.CDLL(path)
    u8p, cp = C.POINTER(C.c_uint8), C.c_char_p
    P = C.POINTER(OrcPacket)
    N = C.POINTER(OrcPiconet)
    sig = {
        "orc_tables_init": (None, []),
        "orc_table": (C.c_int, [cp, C.POINTER(C.c_uint64), C.c_int]),
        "orc_gen_syncword": (C.c_uint64, [C.c_int]),
        "orc_gen_syndrome": (C.c_uint64, [C.c_uint64]),
        "orc_init": (C.c_int, [C.c_int]),
        "orc_reset_syndrome_map": (None, []),
        "orc_syndrome_count": (C.c_uint, []),
        "orc_find_syndrome": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint64)]),
        "orc_find_ac": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), u8p]),
        "orc_find_all": (C.c_size_t, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(OrcHit), C.c_size_t]),
        "orc_unfec13": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "orc_fec23": (C.c_uint16, [C.c_uint16]),
        "orc_unfec23": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
        "orc_unwhiten": (None, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_crcgen": (C.c_uint16, [C.c_void_p, C.c_int, C.c_int]),
        "orc_uap_from_hec": (C.c_uint8, [C.c_uint16, C.c_uint8]),
        "orc_hec_from_uap": (C.c_uint8, [C.c_uint16, C.c_uint8]),
        "orc_packet_new": (P, []),
        "orc_packet_free": (None, [P]),
        "orc_packet_init_found": (None, [P, C.c_uint32, C.c_uint8]),
        "orc_packet_set_data": (None, [P, C.c_void_p, C.c_int, C.c_uint8, C.c_uint32]),
        "orc_packet_set_flag": (None, [P, C.c_int, C.c_int]),
        "orc_packet_get_flag": (C.c_int, [P, C.c_int]),
        "orc_packet_header_packed": (C.c_uint32, [P]),
        "orc_payload_packed": (C.c_int, [P, C.c_void_p]),
        "orc_header_present": (C.c_int, [P]),
        "orc_try_clock": (C.c_uint8, [C.c_int, P]),
        "orc_crc_check": (C.c_int, [C.c_int, P]),
        "orc_decode_header": (C.c_int, [P]),
        "orc_decode_payload": (C.c_int, [P]),
        "orc_decode": (C.c_int, [P]),
        "orc_piconet_new": (N, []),
        "orc_piconet_free": (None, [N]),
        "orc_init_piconet": (None, [N, C.c_uint32]),
        "orc_piconet_set_flag": (None, [N, C.c_int, C.c_int]),
        "orc_piconet_get_flag": (C.c_int, [N, C.c_int]),
        "orc_uap_from_header": (C.c_int, [P, N]),
        "orc_process_packet": (C.c_int, [P, N]),
        "orc_piconet_reset": (None, [N]),
        "orc_perm5": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "orc_hop_precalc": (None, [N]),
        "orc_hop_address_precalc": (None, [C.c_int, N]),
        "orc_gen_hops": (None, [N, C.c_void_p]),
        "orc_get_hop_pattern": (None, [N]),
        "orc_hop_cache_clear": (None, []),
        "orc_single_hop": (C.c_char, [C.c_int, N]),
        "orc_piconet_set_afh_map": (None, [N, C.c_void_p]),
        "orc_init_hop_reversal": (C.c_int, [C.c_int, N]),
        "orc_winnow": (C.c_int, [N]),
        "orc_lap_from_fhs": (C.c_uint32, [P]),
        "orc_uap_from_fhs": (C.c_uint8, [P]),
        "orc_nap_from_fhs": (C.c_uint16, [P]),
        "orc_clock_from_fhs": (C.c_uint32, [P]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    for name in ("orc_fhs", "orc_DM", "orc_DH", "orc_EV3", "orc_EV4", "orc_EV5", "orc_HV"):
        f = getattr(lib, name)
        f.restype, f.argtypes = C.c_int, [C.c_int, P]
    lib.orc_tables_init()
    return lib


@functools.lru_cache(maxsize=None)
def ref():
    """The compiled, unmodified reference (or None when it cannot be had)."""
    path = os.path.join(ORACLE_DIR, "_ref", "libbtbb_ref.so")
    if os.path.isdir("/root/reference/lib/src"):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "btbb_init": (C.c_int, [C.c_int]),
        "btbb_gen_syncword": (C.c_uint64, [C.c_int]),
        "btbb_find_ac": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_int, C.POINTER(vp)]),
        "btbb_packet_new": (vp, []),
        "btbb_packet_unref": (None, [vp]),
        "btbb_packet_set_data": (None, [vp, vp, C.c_int, C.c_uint8, C.c_uint32]),
        "btbb_packet_set_flag": (None, [vp, C.c_int, C.c_int]),
        "btbb_packet_get_flag": (C.c_int, [vp, C.c_int]),
        "btbb_packet_set_uap": (None, [vp, C.c_uint8]),
        "btbb_packet_get_lap": (C.c_uint32, [vp]),
        "btbb_packet_get_uap": (C.c_uint8, [vp]),
        "btbb_packet_get_ac_errors": (C.c_uint8, [vp]),
        "btbb_packet_get_type": (C.c_uint8, [vp]),
        "btbb_packet_get_lt_addr": (C.c_uint8, [vp]),
        "btbb_packet_get_header_flags": (C.c_uint8, [vp]),
        "btbb_packet_get_hec": (C.c_uint8, [vp]),
        "btbb_packet_get_header_packed": (C.c_uint32, [vp]),
        "btbb_packet_get_payload_length": (C.c_int, [vp]),
        "btbb_packet_get_clkn": (C.c_uint32, [vp]),
        "btbb_get_payload_packed": (C.c_int, [vp, vp]),
        "btbb_header_present": (C.c_int, [vp]),
        "btbb_decode_header": (C.c_int, [vp]),
        "btbb_decode_payload": (C.c_int, [vp]),
        "btbb_decode": (C.c_int, [vp]),
        "try_clock": (C.c_uint8, [C.c_int, vp]),
        "crc_check": (C.c_int, [C.c_int, vp]),
        "promiscuous_packet_search": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint8)]),
        "find_known_lap": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint8)]),
        "btbb_piconet_new": (vp, []),
        "btbb_piconet_unref": (None, [vp]),
        "btbb_init_piconet": (None, [vp, C.c_uint32]),
        "btbb_piconet_set_flag": (None, [vp, C.c_int, C.c_int]),
        "btbb_piconet_get_flag": (C.c_int, [vp, C.c_int]),
        "btbb_piconet_get_uap": (C.c_uint8, [vp]),
        "btbb_piconet_set_uap": (None, [vp, C.c_uint8]),
        "btbb_piconet_get_clk_offset": (C.c_int, [vp]),
        "btbb_uap_from_header": (C.c_int, [vp, vp]),
        "btbb_process_packet": (C.c_int, [vp, vp]),
        "refint_find_all": (C.c_size_t, [vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, vp, C.c_size_t]),
        "refint_find_all_mt": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_int, vp, vp, vp, vp, C.c_size_t, vp, vp,
                                         C.POINTER(C.c_double)]),
        "refint_unpack_mt": (C.c_int, [vp, C.c_uint64, vp, C.c_int]),
        "refint_known_lap_chain": (C.c_size_t, [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint8, C.c_uint32,
                                                C.POINTER(C.c_uint64)]),
        "refint_known_lap_chain_records": (C.c_size_t, [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint8, C.c_uint32, vp, C.c_size_t]),
        "refint_clk6_trials": (C.c_uint64, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
        "refint_gen_syndrome": (C.c_uint64, [C.c_uint64]),
        "refint_unfec13": (C.c_int, [vp, vp, C.c_int]),
        "refint_fec23": (C.c_uint16, [C.c_uint16]),
        "refint_unfec23": (C.c_int, [vp, C.c_int, vp]),
        "refint_unwhiten": (None, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "refint_crcgen": (C.c_uint16, [vp, C.c_int, C.c_int]),
        "refint_uap_from_hec": (C.c_uint8, [C.c_uint16, C.c_uint8]),
        "refint_table": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.c_int]),
        "refint_syndrome_count": (C.c_uint, []),
        "refint_find_syndrome": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint64)]),
        "refint_packet_sizeof": (C.c_size_t, []),
        "refint_packet_offsetof": (C.c_size_t, [C.c_char_p]),
        "refint_piconet_candidates": (None, [vp, C.POINTER(C.c_int)]),
        "refint_piconet_packets_observed": (C.c_int, [vp]),
        "refint_piconet_total_packets_observed": (C.c_int, [vp]),
        "refint_piconet_first_pkt_time": (C.c_uint32, [vp]),
        "refint_piconet_flags": (C.c_uint32, [vp]),
        "refint_piconet_sequence": (vp, [vp]),
        "refint_piconet_clock_candidates": (C.POINTER(C.c_uint32), [vp]),
        "refint_piconet_num_candidates": (C.c_int, [vp]),
        "refint_piconet_winnowed": (C.c_int, [vp]),
        "refint_piconet_set_aliased": (None, [vp, C.c_int]),
        "refint_piconet_hop_params": (None, [vp, C.POINTER(C.c_int)]),
        "refint_piconet_observe": (None, [vp, C.c_int, C.c_uint8]),
        "refint_piconet_set_first_pkt_time": (None, [vp, C.c_uint32]),
        "refint_piconet_set_candidate6": (None, [vp, C.c_int, C.c_int]),
        "refint_piconet_set_pattern_index": (None, [vp, C.c_int, C.c_int]),
        "perm5": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "single_hop": (C.c_char, [C.c_int, vp]),
        "get_hop_pattern": (None, [vp]),
        "btbb_piconet_set_afh_map": (None, [vp, vp]),
        "btbb_piconet_set_clk_offset": (None, [vp, C.c_int]),
        "btbb_piconet_set_channel_seen": (C.c_uint8, [vp, C.c_uint8]),
        "btbb_init_hop_reversal": (C.c_int, [C.c_int, vp]),
        "btbb_winnow": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    for name in ("fhs", "DM", "DH", "EV3", "EV4", "EV5", "HV"):
        f = getattr(lib, name)
        f.restype, f.argtypes = C.c_int, [C.c_int, vp]
    return lib


def table(lib, name, cap=256):
    buf = (C.c_uint64 * cap)()
    fn = lib.orc_table if hasattr(lib, "orc_table") else lib.refint_table
    n = fn(name.encode(), buf, cap)
    assert n >= 0, name
    return [int(buf[i]) for i in range(min(n, cap))]


def ptr(a):
    """void* of a contiguous numpy array."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def orc_find_all(stream_u8, search_length, lap, max_err, cap=1 << 20):
    lib = oracle()
    hits = (OrcHit * cap)()
    n = lib.orc_find_all(ptr(stream_u8), search_length, lap, max_err, hits, cap)
    assert n <= cap
    return [(int(h.offset), int(h.lap), int(h.ac_errors)) for h in hits[:n]]


def ref_find_all_native(stream_u8, search_length, lap, max_err, cap=1 << 22, base_offset=0):
    """Same loop as ref_find_all but run in C (oracle/ref_internals.c: refint_find_all)."""
    lib = ref()
    off = np.zeros(cap, np.uint64)
    laps = np.zeros(cap, np.uint32)
    errs = np.zeros(cap, np.uint8)
    n = lib.refint_find_all(C.c_void_p(stream_u8.ctypes.data + base_offset), search_length, lap, max_err,
                            ptr(off), ptr(laps), ptr(errs), cap)
    assert n <= cap
    return [(int(off[i]), int(laps[i]), int(errs[i])) for i in range(n)]


def ref_find_all_mt(stream_u8, search_length, lap, max_err, n_threads, cpus=None, cap_per_thread=None):
    """The all-matches loop on n_threads host threads over disjoint slices of [0, search_length), started and timed
    natively (oracle/ref_internals.c: refint_find_all_mt -- nothing of Python runs between the barrier and the last
    clock read).  Returns (offsets, laps, errors) in stream order as numpy arrays, found per thread, seconds per
    thread, wall seconds."""
    lib = ref()
    bounds = np.linspace(0, search_length, n_threads + 1).astype(np.uint64)
    cap = int(cap_per_thread or (search_length // n_threads) // 1024 + 4096)
    off = np.zeros(n_threads * cap, np.uint64)
    laps = np.zeros(n_threads * cap, np.uint32)
    errs = np.zeros(n_threads * cap, np.uint8)
    found = np.zeros(n_threads, np.uint64)
    secs = np.zeros(n_threads, np.float64)
    wall = C.c_double(0)
    cpu_arr = None if cpus is None else np.ascontiguousarray(np.asarray(cpus, np.int32))
    rc = lib.refint_find_all_mt(ptr(stream_u8), ptr(bounds), n_threads, lap, max_err,
                                None if cpu_arr is None else ptr(cpu_arr), ptr(off), ptr(laps), ptr(errs), cap,
                                ptr(found), ptr(secs), C.byref(wall))
    assert rc == 0
    assert int(found.max()) <= cap, "hit buffer per thread too small"
    keep = np.concatenate([np.arange(i * cap, i * cap + int(found[i])) for i in range(n_threads)]) if n_threads else []
    return off[keep], laps[keep], errs[keep], found, secs, wall.value


def ref_unpack_mt(words_u64, n_threads=16):
    """Packed LSB-first words -> one symbol per byte, natively on n_threads threads."""
    lib = ref()
    words_u64 = np.ascontiguousarray(words_u64, dtype=np.uint64)
    out = np.empty(len(words_u64) * 64, np.uint8)
    lib.refint_unpack_mt(ptr(words_u64), len(words_u64), ptr(out), n_threads)
    return out


def ref_find_all(stream_u8, search_length, lap, max_err):
    """The all-matches loop around the reference's first-match btbb_find_ac."""
    lib = ref()
    out = []
    off = 0
    pkt = C.c_void_p(None)
    base = stream_u8.ctypes.data
    while off < search_length:
        r = lib.btbb_find_ac(C.c_void_p(base + off), int(search_length - off), lap, max_err, C.byref(pkt))
        if r < 0:
            break
        out.append((off + r, int(lib.btbb_packet_get_lap(pkt)), int(lib.btbb_packet_get_ac_errors(pkt))))
        off += r + 1
    if pkt.value:
        lib.btbb_packet_unref(pkt)
    return out


class RefPacketView:
    """Reads fields of a reference btbb_packet through its real offsets."""

    def __init__(self, lib, p):
        self.lib, self.p = lib, p
        self.size = lib.refint_packet_sizeof()

    def _off(self, f):
        o = self.lib.refint_packet_offsetof(f.encode())
        assert o != C.c_size_t(-1).value, f
        return o

    def raw(self):
        return np.ctypeslib.as_array((C.c_uint8 * self.size).from_address(self.p)).copy()

    def field(self, name, dtype, count=1):
        o = self._off(name)
        n = np.dtype(dtype).itemsize * count
        a = np.frombuffer(bytes((C.c_uint8 * n).from_address(self.p + o)), dtype=dtype)
        return a if count > 1 else a[0]
