"""libbtbb_amd -- host-side Python view of the MI355X-native Bluetooth baseband scanner.

The product is the C-ABI shared library ``libbtbb_amd/libbtbb_amd.so`` (SONAME
``libbtbb.so.1``; sources in ``libbtbb_amd/csrc``, headers in ``include/``): a drop-in for the
baseband hot path of libbtbb whose computation runs in hand-written gfx950 HIP kernels.
This package only *binds* it with ctypes for the tests and the benchmark -- there is no
Python or CPU implementation of the path here, and loading fails loudly if the library has
not been built (``python -c "import __graft_entry__ as g; g.build()"``).

PyTorch (when present) is imported *before* the library so that both share one HIP runtime
(torch bundles its own libamdhip64.so); torch tensors' ``data_ptr()`` can then be handed to
the ``btbbx_*_device`` entry points.
"""
import ctypes as C
import os

import numpy as np

try:  # share torch's HIP runtime when torch is around (plumbing only)
    import torch as _torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the C library
    _torch = None

from . import synth  # noqa: F401  (host-side synthetic traffic, numpy)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIBBTBB_AMD_SO") or os.path.join(_HERE, "libbtbb_amd.so")   # the override is for kernel A/B runs

LAP_ANY = 0xFFFFFFFF
PKT_WORDS = 50
MAX_SYMBOLS = 3125

# flag numbers (include/btbb.h)
BTBB_WHITENED, BTBB_NAP_VALID, BTBB_UAP_VALID, BTBB_LAP_VALID = 0, 1, 2, 3
BTBB_CLK6_VALID, BTBB_CLK27_VALID, BTBB_CRC_CORRECT, BTBB_HAS_PAYLOAD = 4, 5, 6, 7
BTBB_GOT_FIRST_PACKET, BTBB_FOLLOWING = 10, 14


class Hit(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("lap", C.c_uint32), ("ac_errors", C.c_uint8),
                ("reserved", C.c_uint8), ("stream", C.c_uint16)]


class Shard(C.Structure):
    _fields_ = [("first_word", C.c_uint64), ("n_words", C.c_uint64), ("search_bits", C.c_uint64),
                ("first_offset", C.c_uint64)]


class Trial(C.Structure):
    _fields_ = [("uap", C.c_uint8), ("type", C.c_uint8), ("rv", C.c_int16)]


class PktIn(C.Structure):
    _fields_ = [("length", C.c_uint32), ("clkn", C.c_uint32), ("flags", C.c_uint32),
                ("uap", C.c_uint8), ("type", C.c_uint8), ("llid", C.c_uint8), ("flow", C.c_uint8)]


class PktOut(C.Structure):
    _fields_ = [("header_rv", C.c_int32), ("payload_rv", C.c_int32), ("payload_length", C.c_int32),
                ("payload_header_length", C.c_int32), ("flags", C.c_uint32), ("header_packed", C.c_uint32),
                ("header_present", C.c_uint8), ("type", C.c_uint8), ("lt_addr", C.c_uint8),
                ("hdr_flags", C.c_uint8), ("hec", C.c_uint8), ("llid", C.c_uint8), ("flow", C.c_uint8),
                ("uap", C.c_uint8), ("payload_header", C.c_uint64), ("payload", C.c_uint64 * 43)]


HIT_DTYPE = np.dtype([("offset", "<u8"), ("lap", "<u4"), ("ac_errors", "u1"), ("reserved", "u1"), ("stream", "<u2")])
TRIAL_DTYPE = np.dtype([("uap", "u1"), ("type", "u1"), ("rv", "<i2")])
PKTIN_DTYPE = np.dtype([("length", "<u4"), ("clkn", "<u4"), ("flags", "<u4"), ("uap", "u1"), ("type", "u1"),
                        ("llid", "u1"), ("flow", "u1")])
PKTOUT_DTYPE = np.dtype([("header_rv", "<i4"), ("payload_rv", "<i4"), ("payload_length", "<i4"),
                         ("payload_header_length", "<i4"), ("flags", "<u4"), ("header_packed", "<u4"),
                         ("header_present", "u1"), ("type", "u1"), ("lt_addr", "u1"), ("hdr_flags", "u1"),
                         ("hec", "u1"), ("llid", "u1"), ("flow", "u1"), ("uap", "u1"),
                         ("payload_header", "<u8"), ("payload", "<u8", (43,))])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 16
assert TRIAL_DTYPE.itemsize == C.sizeof(Trial) == 4
assert PKTIN_DTYPE.itemsize == C.sizeof(PktIn) == 16
assert PKTOUT_DTYPE.itemsize == C.sizeof(PktOut)

_vp, _u64, _u32 = C.c_void_p, C.c_uint64, C.c_uint32

# every symbol include/btbbx.h and include/btbb.h declare: (restype, argtypes)
SIGNATURES = {
    # ---- btbbx.h
    "btbbx_init": (C.c_int, [C.c_int]),
    "btbbx_init_devices": (C.c_int, [_vp, C.c_int, C.c_int]),
    "btbbx_shutdown": (None, []),
    "btbbx_last_error": (C.c_char_p, []),
    "btbbx_device_count": (C.c_int, []),
    "btbbx_table_errors": (C.c_int, []),
    "btbbx_slide_set": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "btbbx_slide_sets_two_level": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "btbbx_malloc": (_vp, [C.c_size_t]),
    "btbbx_free": (None, [_vp]),
    "btbbx_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "btbbx_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "btbbx_memset": (C.c_int, [_vp, C.c_int, C.c_size_t]),
    "btbbx_sync": (C.c_int, [_vp]),
    "btbbx_scan_device": (C.c_int, [_vp, _u64, _u64, _u32, _u64, _u32, C.c_int, _vp, _u32, _vp, _vp]),
    "btbbx_scan_device_fmt": (C.c_int, [_vp, _u64, _u64, _u32, _u64, _u32, C.c_int, C.c_int, _vp, _u32, _vp, _vp]),
    "btbbx_scan_first_device": (C.c_int, [_vp, _u64, _u64, _u32, C.c_int, _vp, _vp]),
    "btbbx_scan_host": (C.c_int64, [_vp, _u64, _u64, _u32, C.c_int, _vp, _u64]),
    "btbbx_scan_symbols": (C.c_int64, [_vp, _u64, _u64, _u32, C.c_int, _vp, _u64]),
    "btbbx_find_first_symbols": (C.c_int, [_vp, _u64, _u64, _u32, C.c_int, _vp]),
    "btbbx_shard_plan": (C.c_int, [_u64, _u32, _u32, _vp]),
    "btbbx_scan_host_multi": (C.c_int64, [_vp, _u64, _u64, _u32, C.c_int, _vp, _u64, _vp, C.c_int]),
    "btbbx_sort_hits": (None, [_vp, C.c_size_t]),
    "btbbx_sort_hits_device": (C.c_int, [_vp, _u32, _vp]),
    "btbbx_order_hits_scratch_bytes": (C.c_size_t, [_u32]),
    "btbbx_scan_ordered_scratch_bytes": (C.c_size_t, [_u64, _u32, _u32, _u32]),
    "btbbx_order_hits_device": (C.c_int, [_vp, _vp, _u32, _vp, C.c_size_t, _vp]),
    "btbbx_order_scan_hits_device": (C.c_int, [_vp, _vp, _u32, _u32, _u64, _vp, C.c_size_t, _vp]),
    "btbbx_scan_ordered_device": (C.c_int, [_vp, _u64, _u64, _u32, _u64, _u32, C.c_int, _vp, _u32, _vp, _vp, C.c_size_t, _vp]),
    "btbbx_scan_ordered_device_fmt": (C.c_int, [_vp, _u64, _u64, _u32, _u64, _u32, C.c_int, C.c_int, _vp, _u32, _vp, _vp, C.c_size_t, _vp]),
    "btbbx_pack_device": (C.c_int, [_vp, _u64, _vp, _vp]),
    "btbbx_unpack_device": (C.c_int, [_vp, _u64, _vp, _vp]),
    "btbbx_msb_to_lsb_device": (C.c_int, [_vp, _u64, _vp]),
    "btbbx_stream_open": (_vp, [_u32, C.c_int, _u64, C.c_int]),
    "btbbx_stream_feed": (C.c_int64, [_vp, _vp, _u64, _vp, _u64]),
    "btbbx_stream_acquire": (_vp, [_vp]),
    "btbbx_stream_submit": (C.c_int64, [_vp, _u64, _vp, _u64]),
    "btbbx_stream_flush": (C.c_int64, [_vp, _vp, _u64]),
    "btbbx_stream_close": (None, [_vp]),
    "btbbx_synth_device": (C.c_int, [_vp, _u64, _u64, _u64, _u32, C.c_int64, _u32, _vp]),
    "btbbx_gather_packets_device": (C.c_int, [_vp, _u64, _u64, _vp, _u32, _u32, _vp, _vp, _vp]),
    "btbbx_trials_device": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "btbbx_decode_device": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "btbbx_decode_hits_device": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "btbbx_decode_hits_counted_device": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "btbbx_decode_hits_piconet_device": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp]),
    "btbbx_decode_hits_piconet_phase_device": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _u32, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "btbbx_uap_table_device": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "btbbx_hop_cfg_init": (None, [_vp, _u32, _vp]),
    "btbbx_hop_sequence_device": (C.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "btbbx_hop_channels_device": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "btbbx_hop_reversal_open": (_vp, [_vp, _u32, C.c_uint8, C.c_int, C.POINTER(C.c_int)]),
    "btbbx_hop_reversal_winnow": (C.c_int, [_vp, _vp, _vp, _u32, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "btbbx_hop_reversal_candidates": (C.c_int64, [_vp, _vp, _u64]),
    "btbbx_hop_reversal_close": (None, [_vp]),
    "btbbx_piconet_state": (C.c_int64, [_vp, C.c_int]),
    "btbbx_piconet_candidates": (C.c_int64, [_vp, _vp, _u64]),
    # ---- btbb.h
    "btbb_init": (C.c_int, [C.c_int]),
    "btbb_get_release": (C.c_char_p, []),
    "btbb_get_version": (C.c_char_p, []),
    "btbb_packet_new": (_vp, []),
    "btbb_packet_ref": (None, [_vp]),
    "btbb_packet_unref": (None, [_vp]),
    "btbb_find_ac": (C.c_int, [_vp, C.c_int, _u32, C.c_int, C.POINTER(_vp)]),
    "btbb_packet_set_flag": (None, [_vp, C.c_int, C.c_int]),
    "btbb_packet_get_flag": (C.c_int, [_vp, C.c_int]),
    "btbb_packet_get_lap": (_u32, [_vp]),
    "btbb_packet_set_uap": (None, [_vp, C.c_uint8]),
    "btbb_packet_get_uap": (C.c_uint8, [_vp]),
    "btbb_packet_get_nap": (C.c_uint16, [_vp]),
    "btbb_packet_set_modulation": (None, [_vp, C.c_uint8]),
    "btbb_packet_set_transport": (None, [_vp, C.c_uint8]),
    "btbb_packet_get_modulation": (C.c_uint8, [_vp]),
    "btbb_packet_get_transport": (C.c_uint8, [_vp]),
    "btbb_packet_get_channel": (C.c_uint8, [_vp]),
    "btbb_packet_get_ac_errors": (C.c_uint8, [_vp]),
    "btbb_packet_get_clkn": (_u32, [_vp]),
    "btbb_packet_get_header_packed": (_u32, [_vp]),
    "btbb_packet_set_data": (None, [_vp, _vp, C.c_int, C.c_uint8, _u32]),
    "btbb_get_symbols": (_vp, [_vp]),
    "btbb_packet_get_payload_length": (C.c_int, [_vp]),
    "btbb_get_payload": (_vp, [_vp]),
    "btbb_get_payload_packed": (C.c_int, [_vp, _vp]),
    "btbb_packet_get_type": (C.c_uint8, [_vp]),
    "btbb_packet_get_lt_addr": (C.c_uint8, [_vp]),
    "btbb_packet_get_header_flags": (C.c_uint8, [_vp]),
    "btbb_packet_get_hec": (C.c_uint8, [_vp]),
    "btbb_gen_syncword": (_u64, [C.c_int]),
    "btbb_decode_header": (C.c_int, [_vp]),
    "btbb_decode_payload": (C.c_int, [_vp]),
    "btbb_print_packet": (None, [_vp]),
    "btbb_header_present": (C.c_int, [_vp]),
    "try_clock": (C.c_uint8, [C.c_int, _vp]),
    "crc_check": (C.c_int, [C.c_int, _vp]),
    "lap_from_fhs": (_u32, [_vp]),
    "uap_from_fhs": (C.c_uint8, [_vp]),
    "nap_from_fhs": (C.c_uint16, [_vp]),
    "clock_from_fhs": (_u32, [_vp]),
    "tun_format": (_vp, [_vp]),
    "btbb_piconet_new": (_vp, []),
    "btbb_piconet_ref": (None, [_vp]),
    "btbb_piconet_unref": (None, [_vp]),
    "btbb_init_piconet": (None, [_vp, _u32]),
    "btbb_piconet_set_uap": (None, [_vp, C.c_uint8]),
    "btbb_piconet_get_uap": (C.c_uint8, [_vp]),
    "btbb_piconet_get_lap": (_u32, [_vp]),
    "btbb_piconet_get_nap": (C.c_uint16, [_vp]),
    "btbb_piconet_get_bdaddr": (_u64, [_vp]),
    "btbb_piconet_get_clk_offset": (C.c_int, [_vp]),
    "btbb_piconet_set_clk_offset": (None, [_vp, C.c_int]),
    "btbb_piconet_set_flag": (None, [_vp, C.c_int, C.c_int]),
    "btbb_piconet_get_flag": (C.c_int, [_vp, C.c_int]),
    "btbb_piconet_set_channel_seen": (C.c_uint8, [_vp, C.c_uint8]),
    "btbb_piconet_clear_channel_seen": (C.c_uint8, [_vp, C.c_uint8]),
    "btbb_piconet_get_channel_seen": (C.c_uint8, [_vp, C.c_uint8]),
    "btbb_piconet_get_afh_map": (_vp, [_vp]),
    "btbb_process_packet": (C.c_int, [_vp, _vp]),
    "btbb_uap_from_header": (C.c_int, [_vp, _vp]),
    "btbb_print_afh_map": (None, [_vp]),
    "btbb_piconet_set_afh_map": (None, [_vp, _vp]),
    "btbb_init_hop_reversal": (C.c_int, [C.c_int, _vp]),
    "btbb_winnow": (C.c_int, [_vp]),
    "btbb_pcapng_create_file": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "btbb_pcapng_append_packet": (C.c_int, [_vp, _u64, C.c_int8, C.c_int8, _u32, C.c_uint8, _vp]),
    "btbb_pcapng_record_bdaddr": (C.c_int, [_vp, _u64, C.c_uint8, C.c_uint8]),
    "btbb_pcapng_record_btclock": (C.c_int, [_vp, _u64, _u64, _u32, _u32]),
    "btbb_pcapng_close": (C.c_int, [_vp]),
    "btbb_pcap_create_file": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "btbb_pcap_append_packet": (C.c_int, [_vp, _u64, C.c_int8, C.c_int8, _u32, C.c_uint8, _vp]),
    "btbb_pcap_close": (C.c_int, [_vp]),
    "btbb_decode": (C.c_int, [_vp]),
    "btbb_init_survey": (C.c_int, []),
    "btbb_next_survey_result": (_vp, []),
}

_lib = None


class BtbbError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library (raises if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BtbbError(
                "%s is missing: build it with `make -C libbtbb_amd/csrc` "
                "(or __graft_entry__.build()); there is no Python/CPU fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)          # RTLD_LOCAL: same symbol names as the reference checker
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError = missing export
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what="btbbx call"):
    if rc < 0:
        raise BtbbError("%s failed (%d): %s" % (what, rc, lib().btbbx_last_error().decode()))
    return rc


def init(max_ac_errors=2):
    """btbb_init() on the current HIP device."""
    check(lib().btbbx_init(max_ac_errors), "btbbx_init")


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------
# thin host-buffer helpers (tests, small jobs); the benchmark uses the *_device entries
# ------------------------------------------------------------------------------------
def init_devices(devices, max_ac_errors=2):
    """btbb_init() on every listed HIP device (for scan_words_multi)."""
    arr = (C.c_int * len(devices))(*devices)
    check(lib().btbbx_init_devices(arr, len(devices), max_ac_errors), "btbbx_init_devices")


def shard_plan(search_bits, n_shards, shard):
    """The library's time-shard plan (btbbx_shard_plan): which words shard `shard` of `n_shards` reads."""
    out = Shard()
    check(lib().btbbx_shard_plan(search_bits, n_shards, shard, C.byref(out)), "btbbx_shard_plan")
    return dict(first_word=out.first_word, n_words=out.n_words, search_bits=out.search_bits,
                first_offset=out.first_offset)


def scan_words(words, search_bits, lap=LAP_ANY, max_ac_errors=2, cap=1 << 20, truncate=False):
    """All access codes in a packed stream held in host memory (numpy uint64).  With truncate=True
    the `cap` smallest (stream, offset) hits are returned when more were found."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    n = check(lib().btbbx_scan_host(_ptr(words), len(words), search_bits, lap, max_ac_errors, _ptr(hits), cap),
              "btbbx_scan_host")
    if n > cap and not truncate:
        raise BtbbError("hit buffer too small: %d > %d" % (n, cap))
    return hits[:min(n, cap)]


def scan_words_multi(words, search_bits, devices, lap=LAP_ANY, max_ac_errors=2, cap=1 << 20, truncate=False):
    """The same through btbbx_scan_host_multi: the capture is time-sharded over `devices` (HIP ordinals,
    repeats allowed), one host thread per entry."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    arr = (C.c_int * len(devices))(*devices)
    n = check(lib().btbbx_scan_host_multi(_ptr(words), len(words), search_bits, lap, max_ac_errors, _ptr(hits), cap,
                                          arr, len(devices)), "btbbx_scan_host_multi")
    if n > cap and not truncate:
        raise BtbbError("hit buffer too small: %d > %d" % (n, cap))
    return hits[:min(n, cap)]


def scan_symbols(symbols, search_length, lap=LAP_ANY, max_ac_errors=2, cap=1 << 20):
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    n = check(lib().btbbx_scan_symbols(_ptr(symbols), len(symbols), search_length, lap, max_ac_errors,
                                       _ptr(hits), cap), "btbbx_scan_symbols")
    if n > cap:
        raise BtbbError("hit buffer too small: %d > %d" % (n, cap))
    return hits[:n]


class DeviceBuffer:
    """A raw HBM allocation owned through the library (no torch needed)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().btbbx_malloc(max(self.nbytes, 8))
        if not self.ptr:
            raise BtbbError("btbbx_malloc(%d): %s" % (nbytes, lib().btbbx_last_error().decode()))

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(lib().btbbx_memcpy_h2d(self.ptr, _ptr(a), a.nbytes), "h2d")
        return self

    def download(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().btbbx_memcpy_d2h(_ptr(out), self.ptr, out.nbytes), "d2h")
        return out

    def zero(self):
        check(lib().btbbx_memset(self.ptr, 0, self.nbytes), "memset")
        return self

    def free(self):
        if self.ptr:
            lib().btbbx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def packets_to_words(symbol_arrays):
    """List of 0/1 symbol arrays -> (n, 50) packed packet words + lengths (host side packing
    of TEST INPUT only; captured streams are packed on the GPU by btbbx_pack_device)."""
    n = len(symbol_arrays)
    words = np.zeros((n, PKT_WORDS), dtype=np.uint64)
    lengths = np.zeros(n, dtype=np.uint32)
    for i, s in enumerate(symbol_arrays):
        s = np.asarray(s, dtype=np.uint8)[:MAX_SYMBOLS]
        lengths[i] = len(s)
        w = synth.pack_bits(s)
        words[i, :len(w)] = w
    return words, lengths


def run_trials(packet_words, pkt_in):
    """64 clock trials per packet on the GPU -> (n, 64) TRIAL_DTYPE."""
    n = len(packet_words)
    d_pk = DeviceBuffer(packet_words.nbytes).upload(packet_words)
    d_in = DeviceBuffer(pkt_in.nbytes).upload(pkt_in)
    d_tr = DeviceBuffer(n * 64 * 4)
    check(lib().btbbx_trials_device(d_pk.ptr, d_in.ptr, n, d_tr.ptr, None), "btbbx_trials_device")
    check(lib().btbbx_sync(None))
    return d_tr.download(TRIAL_DTYPE, n * 64).reshape(n, 64)


def run_uap_table(packet_words, pkt_in=None):
    """n x 64 uint16: try_clock(c) | type << 8 for every packet and CLK1-6 candidate."""
    packet_words = np.ascontiguousarray(packet_words, dtype=np.uint64)
    n = packet_words.shape[0]
    d_pk = DeviceBuffer(packet_words.nbytes).upload(packet_words)
    d_in = DeviceBuffer(pkt_in.nbytes).upload(pkt_in) if pkt_in is not None else None
    d_out = DeviceBuffer(n * 128)
    try:
        check(lib().btbbx_uap_table_device(d_pk.ptr, d_in.ptr if d_in else None, n, d_out.ptr, None), "btbbx_uap_table_device")
        check(lib().btbbx_sync(None), "sync")
        return d_out.download(np.uint16, n * 64).reshape(n, 64)
    finally:
        d_pk.free()
        d_out.free()
        if d_in:
            d_in.free()


def run_decode(packet_words, pkt_in):
    """decode_header + decode_payload per packet on the GPU -> PKTOUT_DTYPE array."""
    n = len(packet_words)
    d_pk = DeviceBuffer(packet_words.nbytes).upload(packet_words)
    d_in = DeviceBuffer(pkt_in.nbytes).upload(pkt_in)
    d_out = DeviceBuffer(n * PKTOUT_DTYPE.itemsize).zero()
    check(lib().btbbx_decode_device(d_pk.ptr, d_in.ptr, n, d_out.ptr, None), "btbbx_decode_device")
    check(lib().btbbx_sync(None))
    return d_out.download(PKTOUT_DTYPE, n)


def run_decode_hits(stream_words, hits, pkt_in, max_length=MAX_SYMBOLS, via_gather=False, init_out=None, count=None):
    """Decode the packets that start at `hits` (HIT_DTYPE: stream, offset) of the packed streams
    stream_words[n_streams, n_words] -> (PKTOUT_DTYPE array, captured lengths).  via_gather=True takes
    the two-step route (btbbx_gather_packets_device + btbbx_decode_device) for comparison.  init_out: what the
    records hold on entry (a PKTOUT_DTYPE array; zeros if None) -- the decoders leave alone what they do not assign.
    count: the list's length as a word in HBM (btbbx_decode_hits_counted_device, capacity len(hits))."""
    stream_words = np.ascontiguousarray(stream_words, dtype=np.uint64)
    n_streams, n_words = stream_words.shape
    n = len(hits)
    d_w = DeviceBuffer(stream_words.nbytes).upload(stream_words)
    d_h = DeviceBuffer(max(hits.nbytes, 16)).upload(hits)
    d_out = DeviceBuffer(n * PKTOUT_DTYPE.itemsize).zero()
    if init_out is not None:
        d_out.upload(np.ascontiguousarray(init_out, dtype=PKTOUT_DTYPE))
    d_len = DeviceBuffer(n * 4).zero()
    pkt_in = np.array(pkt_in, copy=True)
    d_in = d_pk = None
    try:
        if via_gather:
            d_pk = DeviceBuffer(n * PKT_WORDS * 8).zero()
            check(lib().btbbx_gather_packets_device(d_w.ptr, n_words, n_words, d_h.ptr, n, max_length, d_pk.ptr,
                                                    d_len.ptr, None), "btbbx_gather_packets_device")
            check(lib().btbbx_sync(None))
            lengths = d_len.download(np.uint32, n)
            pkt_in["length"] = lengths
            d_in = DeviceBuffer(pkt_in.nbytes).upload(pkt_in)
            check(lib().btbbx_decode_device(d_pk.ptr, d_in.ptr, n, d_out.ptr, None), "btbbx_decode_device")
        else:
            d_in = DeviceBuffer(pkt_in.nbytes).upload(pkt_in)
            if count is not None:
                d_cnt = DeviceBuffer(8).upload(np.array([count, 0], dtype=np.uint32))
                try:
                    check(lib().btbbx_decode_hits_counted_device(d_w.ptr, n_words, n_words, d_h.ptr, d_in.ptr, d_cnt.ptr, n,
                                                                 max_length, d_out.ptr, d_len.ptr, None),
                          "btbbx_decode_hits_counted_device")
                    check(lib().btbbx_sync(None))
                finally:
                    d_cnt.free()
            else:
                check(lib().btbbx_decode_hits_device(d_w.ptr, n_words, n_words, d_h.ptr, d_in.ptr, n, max_length, d_out.ptr,
                                                     d_len.ptr, None), "btbbx_decode_hits_device")
        check(lib().btbbx_sync(None))
        return d_out.download(PKTOUT_DTYPE, n), d_len.download(np.uint32, n)
    finally:
        for b in (d_w, d_h, d_out, d_len, d_in, d_pk):
            if b is not None:
                b.free()


# ---- hop selection / CLK1-27 reversal -------------------------------------------------------
SEQUENCE_LENGTH = 1 << 27


class HopCfg(C.Structure):
    _fields_ = [("address", C.c_uint32), ("afh", C.c_uint8), ("used_channels", C.c_uint8),
                ("reserved", C.c_uint8 * 2), ("bank", C.c_uint8 * 80)]


def hop_cfg(lap, uap, afh_map=None):
    """Kernel configuration of one piconet; afh_map = 10-byte AFH channel map or None."""
    cfg = HopCfg()
    m = None if afh_map is None else np.ascontiguousarray(afh_map, dtype=np.uint8)
    lib().btbbx_hop_cfg_init(C.byref(cfg), ((uap << 24) | lap) & 0xFFFFFFF, None if m is None else _ptr(m))
    return cfg


def hop_sequence(cfg, first=0, count=SEQUENCE_LENGTH):
    """Channels of CLK1-27 values [first, first+count) as a numpy uint8 array (generated in HBM)."""
    buf = DeviceBuffer(count)
    try:
        check(lib().btbbx_hop_sequence_device(C.byref(cfg), first, count, buf.ptr, None), "btbbx_hop_sequence_device")
        check(lib().btbbx_sync(None), "sync")
        return buf.download(np.uint8, count)
    finally:
        buf.free()


def hop_channels(cfg, clocks):
    clocks = np.ascontiguousarray(clocks, dtype=np.uint32)
    d_in, d_out = DeviceBuffer(clocks.nbytes).upload(clocks), DeviceBuffer(len(clocks))
    try:
        check(lib().btbbx_hop_channels_device(C.byref(cfg), d_in.ptr, len(clocks), d_out.ptr, None),
              "btbbx_hop_channels_device")
        check(lib().btbbx_sync(None), "sync")
        return d_out.download(np.uint8, len(clocks))
    finally:
        d_in.free()
        d_out.free()


class HopReversal:
    """Candidate CLK1-27 values of one piconet, held in HBM (btbbx_hop_reversal_*)."""

    def __init__(self, cfg, clk6, channel, aliased=False):
        n = C.c_int(0)
        self.h = lib().btbbx_hop_reversal_open(C.byref(cfg), clk6, channel, int(aliased), C.byref(n))
        if not self.h:
            raise BtbbError("btbbx_hop_reversal_open: %s" % lib().btbbx_last_error().decode())
        self.count = n.value

    def winnow(self, offsets, channels):
        """Apply observed hops in order; returns (stop, count, first candidate)."""
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        ch = np.ascontiguousarray(channels, dtype=np.uint8)
        stop, count, cand0 = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().btbbx_hop_reversal_winnow(self.h, _ptr(off), _ptr(ch), len(off), C.byref(stop), C.byref(count),
                                              C.byref(cand0)), "btbbx_hop_reversal_winnow")
        self.count = count.value
        return stop.value, count.value, cand0.value

    def candidates(self):
        out = np.zeros(max(self.count, 1), np.uint32)
        n = lib().btbbx_hop_reversal_candidates(self.h, _ptr(out), len(out))
        check(min(n, 0), "btbbx_hop_reversal_candidates")
        return out[:n]

    def close(self):
        if self.h:
            lib().btbbx_hop_reversal_close(self.h)
            self.h = None
