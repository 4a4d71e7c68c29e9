"""The C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every symbol include/btbb.h and include/btbbx.h declare.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import libbtbb_amd
    if not os.path.exists(libbtbb_amd.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libbtbb_amd", "csrc")], check=True)
    return libbtbb_amd.lib()


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    names = set(re.findall(r"\b(btbbx?_[a-z0-9_]+)\s*\(", text))
    names |= set(re.findall(r"\b(try_clock|crc_check|[a-z]+_from_fhs|tun_format)\s*\(", text))
    return sorted(names)


@pytest.mark.parametrize("header", ["btbb.h", "btbbx.h"])
def test_every_declared_symbol_is_exported(lib, header):
    import libbtbb_amd
    names = declared(header)
    assert len(names) > 20
    for n in names:
        assert n in libbtbb_amd.SIGNATURES, "no ctypes signature for %s" % n
        assert getattr(lib, n) is not None


def test_nothing_but_the_abi_is_exported():
    """The library takes over libbtbb's SONAME, so it exports exactly its ABI: every defined dynamic symbol is a name one
    of the two headers declares (or a lell_* stub of the reference's header) -- no kernel handles, launchers, table
    builders or template instantiations (-fvisibility=hidden + csrc/exports.map).  Both builds."""
    import libbtbb_amd
    allowed = set(declared("btbb.h")) | set(declared("btbbx.h"))
    for path in (libbtbb_amd.LIB_PATH, os.path.join(os.path.dirname(libbtbb_amd.LIB_PATH), "libbtbb_amd_asan.so")):
        if not os.path.exists(path):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        names = [line.split()[-1] for line in out.splitlines() if line.strip()]
        if "asan" in os.path.basename(path):            # the sanitizer runtime interposes a few names of its own
            names = [n for n in names if not n.startswith(("__asan", "__ubsan", "__sanitizer", "__odr_asan", "__lsan"))]
        extra = [n for n in names if n not in allowed and not n.startswith("lell_")]
        assert not extra, (path, extra[:10])
        assert len([n for n in names if n in allowed]) == len(allowed), sorted(allowed - set(names))[:10]


def test_soname():
    import libbtbb_amd
    out = subprocess.run(["readelf", "-d", libbtbb_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "libbtbb.so.1" in out


def test_host_only_entry_points(lib):
    """Entry points that are pure host bookkeeping work without a GPU."""
    assert lib.btbb_get_version() == b"1.0"
    assert lib.btbb_gen_syncword(0x9E8B33) == 0x4E7A2CCE331A3AE2
    assert lib.btbb_gen_syncword(0x123456) == 0xB048D15A658627C0
    p = lib.btbb_packet_new()
    lib.btbb_packet_set_flag(p, 4, 1)
    assert lib.btbb_packet_get_flag(p, 4) == 1 and lib.btbb_packet_get_flag(p, 2) == 0
    lib.btbb_packet_set_uap(p, 0x47)
    assert lib.btbb_packet_get_uap(p) == 0x47 and lib.btbb_packet_get_flag(p, 2) == 1
    lib.btbb_packet_unref(p)
    pn = lib.btbb_piconet_new()
    lib.btbb_init_piconet(pn, 0xABCDEF)
    assert lib.btbb_piconet_get_lap(pn) == 0xABCDEF and lib.btbb_piconet_get_flag(pn, 3) == 1
    assert lib.btbb_piconet_set_channel_seen(pn, 17) == 1 and lib.btbb_piconet_set_channel_seen(pn, 17) == 0
    lib.btbb_piconet_unref(pn)
    assert lib.btbb_init(9) == -1          # range check happens before any device work


def test_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a device the compute entry points report an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.btbbx_init(2) < 0
    assert b"no CPU path" in lib.btbbx_last_error() or b"HIP" in lib.btbbx_last_error()
    import numpy as np
    import libbtbb_amd
    with pytest.raises(libbtbb_amd.BtbbError):
        libbtbb_amd.scan_words(np.zeros(64, np.uint64), 1000)


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may use oracle/: no product source names it,
    and the shared object neither links it nor has a CPU implementation of the scan to fall back on."""
    import libbtbb_amd
    pkg = os.path.dirname(libbtbb_amd.LIB_PATH)
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text and "orc_" not in text, os.path.join(base, f)
    needed = subprocess.run(["readelf", "-d", libbtbb_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "btbb_ref" not in needed
    # hop selection and capture files, too, refuse to run without a device instead of computing on the host
    import torch
    if not torch.cuda.is_available():
        import ctypes as C
        lib_ = libbtbb_amd.lib()
        cfg = libbtbb_amd.hop_cfg(0x123456, 0x78)
        n = C.c_int(0)
        assert not lib_.btbbx_hop_reversal_open(C.byref(cfg), 0, 0, 0, C.byref(n))
        assert lib_.btbbx_hop_sequence_device(C.byref(cfg), 0, 64, None, None) < 0


def test_headers_are_valid_c90_and_cxx(tmp_path):
    """Both public headers compile stand-alone as pedantic C90 and as C++ (plain C ABI, no torch types)."""
    src = tmp_path / "h.c"
    src.write_text("#include <btbb.h>\n#include <btbbx.h>\nint main(void) { return 0; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c90", "-pedantic", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")], check=True)
    text = open(os.path.join(inc, "btbbx.h")).read() + open(os.path.join(inc, "btbb.h")).read()
    assert "torch" not in text and "hip/" not in text


def test_le_symbols_bind_and_fail_loudly():
    """The LE half of libbtbb's ABI (lell_*, reference btbb.h:229-281) is outside this library, but the
    SONAME is libbtbb's: every LE symbol is exported so that a program linked against the reference binds,
    and calling one aborts at once with a diagnostic instead of dying in a lazy symbol lookup."""
    import libbtbb_amd
    out = subprocess.run(["nm", "-D", "--defined-only", libbtbb_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    have = {line.split()[-1] for line in out.splitlines() if " T " in line}
    want = {"lell_allocate_and_decode", "lell_packet_new", "lell_packet_ref", "lell_packet_unref",
            "lell_get_access_address", "lell_get_access_address_offenses", "lell_packet_is_data",
            "lell_get_channel_index", "lell_get_channel_k", "lell_get_adv_type_str", "lell_print",
            "lell_pcapng_create_file", "lell_pcapng_append_packet", "lell_pcapng_record_connect_req",
            "lell_pcapng_close", "lell_pcap_create_file", "lell_pcap_ppi_create_file", "lell_pcap_append_packet",
            "lell_pcap_append_ppi_packet", "lell_pcap_close"}
    assert want <= have
    code = ("import ctypes, sys; sys.path.insert(0, %r); import libbtbb_amd as bt; "
            "bt.lib(); ctypes.CDLL(bt.LIB_PATH).lell_packet_new()" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "lell_packet_new" in r.stderr and "not" in r.stderr
