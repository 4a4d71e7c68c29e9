"""A plain C90 caller compiled against include/btbb.h and linked with the drop-in library:
the CPU part proves the header is valid C and every symbol the caller uses links; the GPU part
runs it and compares with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _libs
from libbtbb_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "dropin_caller.c")


def build(tmp_path):
    import libbtbb_amd
    if not os.path.exists(libbtbb_amd.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libbtbb_amd", "csrc")], check=True)
    exe = str(tmp_path / "dropin_caller")
    libdir = os.path.dirname(libbtbb_amd.LIB_PATH)
    cmd = ["gcc", "-std=c90", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", exe, "-L", libdir, "-l:libbtbb_amd.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True)
    return exe


def test_c90_caller_compiles_and_links(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libbtbb.so.1" in out            # linked by SONAME, as against the reference library


@pytest.mark.gpu
def test_c90_caller_runs_like_the_oracle(tmp_path):
    exe = build(tmp_path)
    rng = np.random.default_rng(5)
    lap, uap = 0x4D5A11, 0x6B
    sym = rng.integers(0, 2, 40000, dtype=np.uint8)
    pos, clk = 300, 1000
    placed = []
    for k in range(9):
        t = [synth.TYPE_NULL, synth.TYPE_POLL, synth.TYPE_DM1, synth.TYPE_DH1][k % 4]
        clk6 = (pos // 312 >> 1) & 63          # the caller passes clkn = offset / 312
        p = synth.build_packet(lap, uap, clk6, t, lt_addr=2, body=bytes(range(k + 3)))
        sym[pos:pos + len(p)] = p
        placed.append(pos)
        pos += 4000
    path = str(tmp_path / "capture.sym")
    sym.tofile(path)
    # the executable asks for the SONAME libbtbb.so.1, exactly as if linked against libbtbb
    libdir = tmp_path / "lib"
    libdir.mkdir()
    os.symlink(sys.modules["libbtbb_amd"].LIB_PATH, str(libdir / "libbtbb.so.1"))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = str(libdir) + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    prefix = str(tmp_path / "cap")
    res = subprocess.run([exe, path, hex(lap), prefix], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("AC ")]
    orc = _libs.oracle()
    orc.orc_init(2)
    want = _libs.orc_find_all(np.ascontiguousarray(sym), len(sym) - 63, _libs.LAP_ANY, 2)
    got = [(int(l.split()[1].split("=")[1]), int(l.split()[2].split("=")[1], 16), int(l.split()[3].split("=")[1])) for l in lines]
    assert got == want
    assert [g[0] for g in got if g[1] == lap] == placed
    done = [l for l in res.stdout.splitlines() if l.startswith("DONE")][0]
    assert "found=%d" % len(want) in done and "uap_valid=1" in done and "uap=%02x" % uap in done
    # the capture files it wrote: one record per access code, LINKTYPE_BLUETOOTH_BREDR_BB pseudo header
    import struct
    import _capture
    recs = _capture.pcap_records(open(prefix + ".pcap", "rb").read())
    assert len(recs) == len(want)
    for (sec, nsec, rec), (o, l, e) in zip(recs, want):
        ch, sig, noise, errs = struct.unpack_from("<BbbB", rec, 0)
        rec_lap, ref = struct.unpack_from("<II", rec, 8)
        assert (ch, sig, noise, errs, rec_lap) == (17, -40, -90, e, l) and ref == (lap | 0xFF << 24)
        assert sec * 10**9 + nsec == o * 1000
    blocks = _capture.pcapng_blocks(_capture.normalize_pcapng(open(prefix + ".pcapng", "rb").read()))
    assert [b[0] for b in blocks] == [0x0A0D0D0A, 1] + [6] * len(want)
    assert b"dropin_caller" in blocks[1][1] and struct.pack("<HH", 0xD340, 12) in blocks[1][1]
