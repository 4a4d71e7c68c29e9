"""The occupancy every measured number in DESIGN.md rests on, read from the code objects inside the built library:
registers, LDS and scratch of the hot kernels (the AMDGPU metadata note of each gfx950 code object in .hip_fatbin).
A change that pushes a kernel over a register or LDS step -- or makes a decoder state live in scratch -- shows up here,
on the CPU, before a GPU run is spent on it."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "libbtbb_amd", "libbtbb_amd.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
LDS_PER_CU = 160 * 1024
VGPRS_PER_SIMD_LANE = 512


def _kernels():
    data = open(SO, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = {}, 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                path = "/tmp/btbb_kres_%d_%d.co" % (os.getpid(), len(out))
                with open(path, "wb") as f:
                    f.write(data[i + off:i + off + size])
                notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
                os.unlink(path)
                for block in notes.split("- .agpr_count:")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", block).group(1)
                    out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                                 for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                           "group_segment_fixed_size", "max_flat_workgroup_size")}
        pos = i + 24
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(SO) or not os.path.exists(READELF):
        pytest.skip("library or llvm-readelf missing")
    k = _kernels()
    assert len(k) > 30, sorted(k)
    return k


def _one(kernels, pattern):
    m = [n for n in kernels if re.search(pattern, n)]
    assert len(m) == 1, (pattern, m)
    return kernels[m[0]]


def _waves_per_simd(vgprs):
    return min(8, VGPRS_PER_SIMD_LANE // (-(-vgprs // 8) * 8))


def test_lap_any_kernel_two_workgroups_of_768_per_cu(kernels):
    """2 x 768 threads per CU = 6 waves per SIMD; the set + rings are dynamic LDS (76 KiB per workgroup)."""
    # (MSB first or not) x (hits appended to a list, or left in their segment's slots for the ordered scan: round 6)
    for pat in (r"scan_slide_kernelI8SlideStdLi2ELb0ELb0E", r"scan_slide_kernelI8SlideStdLi2ELb1ELb0E",
                r"scan_slide_kernelI8SlideStdLi2ELb0ELb1E", r"scan_slide_kernelI8SlideStdLi2ELb1ELb1E"):
        k = _one(kernels, pat)
        assert k["vgpr_count"] <= 80 and _waves_per_simd(k["vgpr_count"]) >= 6, k
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        assert k["max_flat_workgroup_size"] == 768, k


def test_lap_any_kernel_for_three_and_four_errors_one_workgroup_per_cu(kernels):
    """the two-level form owns the CU -- 1024 threads = 4 waves per SIMD, the 2^20-bit set + rings = the whole LDS."""
    for msb in (0, 1):
        k = _one(kernels, r"scan_slide_kernelI6Slide4Li3ELb%dELb0E" % msb)
        assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        assert k["max_flat_workgroup_size"] == 1024, k


def test_known_lap_kernel_eight_waves_per_simd(kernels):
    for cls in (0, 1):
        k = _one(kernels, r"scan_known_lap_kernelILi2ELi%dELb0ELb0E" % cls)
        assert k["vgpr_count"] <= 64 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        assert k["group_segment_fixed_size"] * 7 <= LDS_PER_CU, k
        # the ordered scan's form (round 6: hits leave through the segment slots): seven waves per SIMD, nothing in scratch
        k = _one(kernels, r"scan_known_lap_kernelILi2ELi%dELb0ELb1E" % cls)
        assert _waves_per_simd(k["vgpr_count"]) >= 7 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        assert k["group_segment_fixed_size"] * 7 <= LDS_PER_CU, k


def test_decode_hits_kernel_six_workgroups_per_cu(kernels):
    """DESIGN 3.4 / NOTEBOOK 3.4: 256 threads, six workgroups per CU by LDS (stage + result copies + tables), six waves per SIMD by
    registers; the decoder state lives in registers (a few dwords of spill are tolerated, a PState in scratch is not)."""
    k = _one(kernels, r"decode_hits_kernel")
    assert k["vgpr_count"] <= 80, k
    assert k["group_segment_fixed_size"] * 6 <= LDS_PER_CU, k
    assert k["private_segment_fixed_size"] <= 32 and k["vgpr_spill_count"] <= 8, k
    # (round 4: DM / DH / EV4 / EV5 payloads beyond 256 bits are decoded by a group of lanes per packet at the end of this kernel;
    # a careless version of that loop made it spill 22 - 40 registers and the single-slot mix lost 20 %: profiles/r04_decode)


def test_decoders_and_trials_keep_their_state_in_registers(kernels):
    for pat in (r"^_Z13decode_kernel", r"decode_bytes_kernel", r"trials_wide_kernel", r"trials_state_kernel", r"trials_merge_kernel",
                r"replay_kernel"):
        k = _one(kernels, pat)
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (pat, k)
    k = _one(kernels, r"trials_linear_kernel")              # two 512-thread workgroups per CU (round 5): four waves per SIMD at
    assert k["vgpr_count"] <= 128, k                        # the 128-register ceiling, nothing in scratch, half the LDS each
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and 2 * k["group_segment_fixed_size"] <= LDS_PER_CU, k
    assert k["max_flat_workgroup_size"] == 512, k


def test_order_kernels_have_no_scratch(kernels):
    """(order_crowded_kernel, the cold pass over buckets of thousands, keeps one record in private memory -- and so does
    order_single_kernel, the segment slots' fallback for streams of sync words, which contains it)"""
    for name, k in kernels.items():
        if "order_" in name and "crowded" not in name and "order_single" not in name:
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (name, k)
