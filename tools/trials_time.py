#!/usr/bin/env python3
"""Throughput of the 64-clock trial kernel and of the batch decoder on random packets of every type
(run on the MI355X box)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import libbtbb_amd as bt
import _pkt

bt.init(2)
lib = bt.lib()
rng = np.random.default_rng(77)
pk = _pkt.random_packets(rng, 1024)
syms = [np.ascontiguousarray(s[:bt.MAX_SYMBOLS]) for s, _ in pk]
words, lengths = bt.packets_to_words(syms)
reps = 128
n = len(syms) * reps
pin = np.zeros(len(syms), bt.PKTIN_DTYPE)
pin["length"] = lengths
pin["flags"] = 1
d_pk = torch.from_numpy(np.tile(words.view(np.int64), (reps, 1))).cuda()
d_in = torch.from_numpy(np.tile(pin, reps).view(np.uint8)).cuda()
d_tr = torch.zeros(n * 64, dtype=torch.int32, device="cuda")
hs = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    bt.check(lib.btbbx_trials_device(d_pk.data_ptr(), d_in.data_ptr(), n, d_tr.data_ptr(), hs))


run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
print("trials: %d packets (all 16 types, random lengths) in %.3f ms = %.1f M packets/s, checksum %d"
      % (n, ms, n / ms / 1e3, int(d_tr.sum().item())))

# small batches: latency shape
for k in (1, 16, 128, 129, 1024):
    def small():
        bt.check(lib.btbbx_trials_device(d_pk.data_ptr(), d_in.data_ptr(), k, d_tr.data_ptr(), hs))
    small()
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        small()
    b.record()
    torch.cuda.synchronize()
    print("trials, %4d packets: %.1f us per launch" % (k, a.elapsed_time(b) / 20 * 1e3))
