"""The host side of the drop-in under AddressSanitizer + UBSan (SURVEY.md 5, row "sanitizers").

`make -C libbtbb_amd/csrc asan` compiles every .cpp of the library -- packet / piconet objects with refcounts and interior
pointers (btbb_api.cpp, piconet.cpp), the capture-file writers (capture.cpp), the streaming ingest (stream.cpp), contexts
and per-call leases (context.cpp) -- with g++ -fsanitize=address,undefined and links them with the normal device objects.
The tests below re-run product tests in a child interpreter that loads that library (LD_PRELOAD of the sanitizer run
times); any report aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_SO = os.path.join(ROOT, "libbtbb_amd", "libbtbb_amd_asan.so")


def runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def run_under_sanitizers(test_args, timeout=1200):
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer run times not installed")
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):      # (re)build where the compiler is; the file travels to the GPU box
        r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libbtbb_amd", "csrc"), "asan"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(ASAN_SO)
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, LIBBTBB_AMD_SO=ASAN_SO,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + test_args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    text = r.stdout + r.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-6000:]
    assert r.returncode == 0, text[-6000:]
    return text


def test_host_side_cpu_tests_under_asan_ubsan():
    """capture-file writers (byte-identical files vs the compiled reference), ABI / loud-failure tests, the candidate set"""
    out = run_under_sanitizers(["tests/test_capture_files.py", "tests/test_abi_exports.py", "tests/test_slide_checks.py", "-m", "not gpu"])
    assert " passed" in out


@pytest.mark.gpu
def test_drop_in_and_piconet_sequences_under_asan_ubsan():
    """the drop-in on the GPU: packet objects through find / set_data / decode, btbb_uap_from_header / btbb_process_packet
    ladders (piconet.cpp), capture files of GPU-decoded packets, the streaming ingest"""
    out = run_under_sanitizers(["tests/test_gpu_capture.py", "tests/test_gpu_hop.py", "tests/test_gpu_packets.py",
                                "tests/test_gpu_scan.py", "-m", "gpu", "-k",
                                "(drop_in or process_packet or capture or streaming_ingest or piconet or winnow) and not concurrent"])   # (tests that
    # create torch tensors are left out: torch's lazy CUDA initialisation does not survive LD_PRELOAD of the sanitizer run times)
    assert " passed" in out
