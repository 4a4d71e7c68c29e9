"""pcap / pcapng writers (libbtbb_amd/csrc/capture.cpp) against the compiled reference
(lib/src/pcap.c, pcapng.c, pcapng-bt.c): same files byte for byte, apart from the bytes the
reference leaves undefined.  Packets here carry no payload (that needs the GPU decode; see
tests/test_gpu_capture.py) -- file structure, options, records, error codes."""
import ctypes as C
import os

import numpy as np
import pytest

import _capture
import _libs
import libbtbb_amd as bt

ref = _libs.ref()
pytestmark = pytest.mark.skipif(ref is None or not hasattr(ref, "btbb_pcapng_create_file"),
                                reason="compiled reference (oracle/_ref) not available")
vp = C.c_void_p
APPEND = [vp, C.c_uint64, C.c_int8, C.c_int8, C.c_uint32, C.c_uint8, vp]


@pytest.fixture(scope="module")
def libs():
    for name, res, args in (("btbb_pcapng_create_file", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp)]),
                            ("btbb_pcapng_append_packet", C.c_int, APPEND),
                            ("btbb_pcapng_record_bdaddr", C.c_int, [vp, C.c_uint64, C.c_uint8, C.c_uint8]),
                            ("btbb_pcapng_record_btclock", C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
                            ("btbb_pcapng_close", C.c_int, [vp]),
                            ("btbb_pcap_create_file", C.c_int, [C.c_char_p, C.POINTER(vp)]),
                            ("btbb_pcap_append_packet", C.c_int, APPEND),
                            ("btbb_pcap_close", C.c_int, [vp]),
                            ("btbb_packet_set_transport", None, [vp, C.c_uint8]),
                            ("btbb_packet_set_modulation", None, [vp, C.c_uint8])):
        f = getattr(ref, name)
        f.restype, f.argtypes = res, args
    return bt.lib(), ref


def _packets(lib, rng, n):
    """The same header-only packets in one library."""
    out = []
    for i in range(n):
        p = vp(lib.btbb_packet_new())
        sym = rng.integers(0, 2, 200, dtype=np.uint8)
        lib.btbb_packet_set_data(p, _libs.ptr(sym), len(sym), int(rng.integers(0, 79)), int(rng.integers(0, 1 << 28)))
        lib.btbb_packet_set_transport(p, int(rng.integers(0, 5)))
        lib.btbb_packet_set_modulation(p, int(rng.integers(0, 3)))
        out.append(p)
    return out


def _script(lib, path, kind, desc):
    """Write one capture file with `lib`; returns the list of return codes."""
    rng = np.random.default_rng(808)
    rcs = []
    h = vp()
    if kind == "pcapng":
        rcs.append(lib.btbb_pcapng_create_file(path.encode(), desc, C.byref(h)))
    else:
        rcs.append(lib.btbb_pcap_create_file(path.encode(), C.byref(h)))
    append = lib.btbb_pcapng_append_packet if kind == "pcapng" else lib.btbb_pcap_append_packet
    pk = _packets(lib, rng, 9)
    ns = 1_700_000_000_123_456_789
    for i, p in enumerate(pk):
        sig, noise = int(rng.integers(-90, -20)), int(rng.integers(-100, -10))
        reflap = 0xFFFFFFFF if i % 3 == 0 else int(rng.integers(0, 1 << 24))
        refuap = 0xFF if i % 4 == 0 else int(rng.integers(0, 255))
        rcs.append(append(h, ns, sig, noise, reflap, refuap, p))
        ns += int(rng.integers(1, 1 << 33))
        if kind == "pcapng" and i == 2:
            rcs.append(lib.btbb_pcapng_record_bdaddr(h, 0x0000A1B2C3D4E5F6, 0xFF, 1))
        if kind == "pcapng" and i == 5:
            rcs.append(lib.btbb_pcapng_record_btclock(h, 0x0000A1B2C3D4E5F6, ns, 0x0ABCDEF, 0x0FFFFFFF))
            rcs.append(lib.btbb_pcapng_record_bdaddr(h, 0x123456789ABC, 0x0F, 0))
    rcs.append(lib.btbb_pcapng_close(h) if kind == "pcapng" else lib.btbb_pcap_close(h))
    for p in pk:
        lib.btbb_packet_unref(p)
    return rcs


@pytest.mark.parametrize("desc", [None, b"", b"MI355X scanner", b"x" * 300])
def test_pcapng_files_equal_reference(libs, tmp_path, desc):
    lib, r = libs
    a, b = str(tmp_path / "ours.pcapng"), str(tmp_path / "ref.pcapng")
    assert _script(lib, a, "pcapng", desc) == _script(r, b, "pcapng", desc)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) == len(db)
    assert _capture.normalize_pcapng(da) == _capture.normalize_pcapng(db)
    assert _capture.normalize_pcapng(da) == da            # ours has zeros where the reference has garbage
    kinds = [t for t, _ in _capture.pcapng_blocks(da)]
    assert kinds == [0x0A0D0D0A, 1] + [6] * 9


def test_pcap_files_equal_reference(libs, tmp_path):
    lib, r = libs
    a, b = str(tmp_path / "ours.pcap"), str(tmp_path / "ref.pcap")
    assert _script(lib, a, "pcap", None) == _script(r, b, "pcap", None)
    da = open(a, "rb").read()
    assert da == open(b, "rb").read()
    recs = _capture.pcap_records(da)
    assert len(recs) == 9 and all(len(x[2]) == 22 for x in recs)


def test_error_codes(libs, tmp_path):
    lib, r = libs
    for L in (lib, r):
        path = str(tmp_path / ("exists_%d.pcapng" % id(L)))
        h = vp()
        assert L.btbb_pcapng_create_file(path.encode(), None, C.byref(h)) == 0
        assert L.btbb_pcapng_close(h) == -1                       # always "invalid handle" (pcapng-bt.c:335-343)
        h2 = vp()
        # O_EXCL: the file exists; every open() failure surfaces as FILE_WRITE_ERROR (pcapng.c:100-102)
        assert L.btbb_pcapng_create_file(path.encode(), None, C.byref(h2)) == -6
        assert L.btbb_pcapng_create_file(str(tmp_path / "no/such/dir/f").encode(), None, C.byref(h2)) == -6
        assert L.btbb_pcap_create_file(str(tmp_path / "no/such/dir/f").encode(), C.byref(h2)) == -2
        assert L.btbb_pcap_close(None) == -1
        assert L.btbb_pcap_append_packet(None, 0, 0, 0, 0, 0, None) == -1
