import os, sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import libbtbb_amd as bt
from libbtbb_amd import synth
bt.init(2); lib = bt.lib()
words, _ = synth.make_stream(5, 2048, stride=4096)
sym = np.ascontiguousarray(synth.unpack_bits(words))
for n in (4096, 65536):
    small = np.ascontiguousarray(sym[: n + 72]); pkt = C.c_void_p(None)
    for _ in range(5): r = lib.btbb_find_ac(small.ctypes.data, n, bt.LAP_ANY, 2, C.byref(pkt))
    t0 = time.perf_counter()
    for _ in range(200): r = lib.btbb_find_ac(small.ctypes.data, n, bt.LAP_ANY, 2, C.byref(pkt))
    print(n, "symbols: find_ac us", round((time.perf_counter() - t0) / 200 * 1e6, 1), "offset", r)
# a caller walking a buffer (search, take the match, search again one symbol further): microseconds per call
for n, stride in ((65536, 4096), (8192, 1024)):
    w2, _ = synth.make_stream(9, n // 64 + 2, stride=stride)
    buf = np.ascontiguousarray(synth.unpack_bits(w2)[: n + 63]); pkt = C.c_void_p(None)
    def walk():
        off, calls = 0, 0
        while True:
            left = len(buf) - 63 - off
            if left <= 0:
                break
            r = lib.btbb_find_ac(buf.ctypes.data + off, left, bt.LAP_ANY, 2, C.byref(pkt))
            calls += 1
            if r < 0:
                break
            off += r + 1
        return calls
    walk()
    t0 = time.perf_counter(); calls = sum(walk() for _ in range(20)); dt = time.perf_counter() - t0
    print(n, "symbols walked:", calls // 20, "calls per walk,", round(dt / calls * 1e6, 1), "us per call (BTBB_FIND_AC_WINDOW=%s)" % os.environ.get("BTBB_FIND_AC_WINDOW", "1"))
# drop-in decode and UAP discovery round trips
from libbtbb_amd import synth as _s
rng = np.random.default_rng(1)
symp = _s.build_packet(0x123456, 0x9A, 17, 10, lt_addr=1, body=bytes(range(100)), fhs_bits=_s.fhs_payload(0x123456, 0x9A, 1, 2, rng))
symp = np.ascontiguousarray(np.concatenate([symp, rng.integers(0, 2, 40, dtype=np.uint8)]))
p = C.c_void_p(lib.btbb_packet_new())
lib.btbb_packet_set_flag(p, 0, 1)
lib.btbb_packet_set_data(p, symp.ctypes.data, len(symp), 3, 17 << 1)
lib.btbb_packet_set_uap(p, 0x9A)
lib.btbb_packet_set_flag(p, 4, 1)
import os
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
for _ in range(5): rv = lib.btbb_decode(p)
t0 = time.perf_counter()
for _ in range(200): rv = lib.btbb_decode(p)
dt = (time.perf_counter() - t0) / 200 * 1e6
pn = C.c_void_p(lib.btbb_piconet_new()); lib.btbb_init_piconet(pn, 0x123456)
t0 = time.perf_counter()
for _ in range(100):
    lib.btbb_piconet_set_flag(pn, 10, 0); lib.btbb_piconet_set_flag(pn, 2, 0); lib.btbb_piconet_set_flag(pn, 4, 0)
    lib.btbb_uap_from_header(p, pn)
du = (time.perf_counter() - t0) / 100 * 1e6
C.CDLL(None).fflush(None)
os.dup2(saved, 1)
print("btbb_decode (DM3, 100 bytes) us", round(dt, 1), "rv", rv, "| btbb_uap_from_header us", round(du, 1))
