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


def _pc_flags(prefix):
    """`pkg-config --cflags --libs libbtbb` for the scratch prefix -- through pkg-config when the box
    has one, otherwise by expanding the installed libbtbb.pc the same way."""
    pcdir = os.path.join(prefix, "lib", "pkgconfig")
    import shutil
    if shutil.which("pkg-config"):
        env = dict(os.environ, PKG_CONFIG_PATH=pcdir)
        return subprocess.run(["pkg-config", "--cflags", "--libs", "libbtbb"], check=True, capture_output=True,
                              text=True, env=env).stdout.split()
    var, fields = {}, {}
    for line in open(os.path.join(pcdir, "libbtbb.pc")):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if ":" in line and ("=" not in line or line.index(":") < line.index("=")):
            k, v = line.split(":", 1)
            fields[k.strip()] = v.strip()
        else:
            k, v = line.split("=", 1)
            var[k.strip()] = v.strip()

    def expand(v):
        for _ in range(5):
            for k, val in var.items():
                v = v.replace("${%s}" % k, val)
        return v
    assert fields["Name"].startswith("libbtbb") and fields["Version"]
    return (expand(fields["Cflags"]) + " " + expand(fields["Libs"])).split()


def _install(tmp_path):
    prefix = str(tmp_path / "prefix")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libbtbb_amd", "csrc"), "install", "PREFIX=" + prefix],
                   check=True, capture_output=True)
    return prefix


def test_install_layout_and_pkg_config_build(tmp_path):
    """`make install` lays the library out like the reference's cmake install (libbtbb.so.1.0, SONAME
    and dev links, btbb.h, libbtbb.pc) and the C90 caller builds from the installed files alone with
    the flags libbtbb.pc gives (lib/libbtbb.pc.in, lib/src/CMakeLists.txt:42-67)."""
    prefix = _install(tmp_path)
    lib = os.path.join(prefix, "lib")
    assert os.path.isfile(os.path.join(lib, "libbtbb.so.1.0")) and not os.path.islink(os.path.join(lib, "libbtbb.so.1.0"))
    assert os.readlink(os.path.join(lib, "libbtbb.so.1")) == "libbtbb.so.1.0"
    assert os.readlink(os.path.join(lib, "libbtbb.so")) == "libbtbb.so.1"
    assert os.path.isfile(os.path.join(prefix, "include", "btbb.h")) and os.path.isfile(os.path.join(prefix, "include", "btbbx.h"))
    flags = _pc_flags(prefix)
    assert "-lbtbb" in flags and any(f.startswith("-I") for f in flags)
    exe = str(tmp_path / "caller_pc")
    subprocess.run(["gcc", "-std=c90", "-pedantic", "-Wall", "-Werror", SRC, "-o", exe] + flags +
                   ["-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libbtbb.so.1" in out and "libbtbb_amd" not in out
    out = subprocess.run(["readelf", "-d", os.path.join(lib, "libbtbb.so.1.0")], capture_output=True, text=True).stdout
    assert "SONAME" in out and "libbtbb.so.1" in out


def _sketch_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    block = sec[sec.index("```c") + 4:]
    return block[:block.index("```")]


def _build_sketch(tmp_path):
    prefix = _install(tmp_path)
    (tmp_path / "sketch_block.inc").write_text(_sketch_block())
    exe = str(tmp_path / "sketch")
    lib = os.path.join(prefix, "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(tmp_path),
                    os.path.join(ROOT, "tests", "c", "sketch_harness.c"), "-o", exe] + _pc_flags(prefix) +
                   ["-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True)
    return exe


def test_integration_sketch_compiles(tmp_path):
    """The reference-side patch printed in INTEGRATION.md section 2 is real code: it compiles (with the
    three reference-internal names it uses stood in by the harness) and links against the install."""
    assert "btbbx_find_first_symbols" in _sketch_block()
    _build_sketch(tmp_path)


@pytest.mark.gpu
def test_integration_sketch_first_match(tmp_path):
    """...and returns the FIRST access code of a window that holds several (first-match semantics of
    btbb_find_ac), call after call, exactly like the shipped drop-in and the oracle."""
    exe = _build_sketch(tmp_path)
    rng = np.random.default_rng(11)
    sym = rng.integers(0, 2, 30000, dtype=np.uint8)
    for k, pos in enumerate((700, 701 + 64, 5000, 5200, 12345, 29000)):       # dense and sparse neighbours
        sym[pos:pos + 64] = synth.bits_lsb(synth.syncword(0x100000 + 77 * k), 64)
    path = str(tmp_path / "win.sym")
    sym.tofile(path)
    res = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    got = [int(l.split("=")[1]) for l in res.stdout.splitlines() if l.startswith("AC ")]
    orc = _libs.oracle()
    orc.orc_init(2)
    want = [o for (o, l, e) in _libs.orc_find_all(np.ascontiguousarray(sym), len(sym) - 63, _libs.LAP_ANY, 2)]
    assert got == want and len(got) >= 6
    assert "bad=0" in res.stdout
