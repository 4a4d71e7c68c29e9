#!/usr/bin/env python3
"""How many records of the config-3 capture share their ordering bucket, and what the ordering costs (GPU box):
  python tools/order_probe.py
Scan alone (btbbx_scan_device) against scan + order (btbbx_scan_ordered_device) with HIP events; then the `work` words
order_scatter_kernel left in the scratch (~0 = alone in its bucket)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import libbtbb_amd as bt  # noqa: E402
from libbtbb_amd import synth  # noqa: E402


def main():
    bt.init(2)
    lib = bt.lib()
    dev = torch.device("cuda:0")
    hs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lap, uap = 0x9E8B33, 0x47
    nch, wpc0, tiles = 79, 1 << 14, 64
    rng = np.random.default_rng(7)
    base = synth.noise_words(10, 0, nch * wpc0).reshape(nch, wpc0)
    slots = wpc0 * 64 // 4096 - 1
    for ch in range(nch):
        symc = synth.unpack_bits(base[ch])
        for k in range(slots):
            p = synth.build_packet(lap, uap, k & 63, synth.TYPE_DM1, lt_addr=1 + k % 7, body=b"abcd")
            pos = k * 4096 + 100 + int(rng.integers(0, 64))
            symc[pos:pos + len(p)] = p
        base[ch] = synth.pack_bits(symc)
    wpc = wpc0 * tiles
    d3 = torch.from_numpy(base.view(np.int64)).to(dev).repeat(1, tiles).contiguous()
    nbits = wpc * 64 - 63
    cap = nch * slots * tiles + (1 << 16)
    hits = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ob = lib.btbbx_order_hits_scratch_bytes(cap)
    scratch = torch.empty(ob, dtype=torch.uint8, device=dev)

    def plain():
        cnt.zero_()
        bt.check(lib.btbbx_scan_device(d3.data_ptr(), wpc, wpc, nch, nbits, lap, 2, hits.data_ptr(), cap, cnt.data_ptr(), hs))

    def ordered():
        cnt.zero_()
        bt.check(lib.btbbx_scan_ordered_device(d3.data_ptr(), wpc, wpc, nch, nbits, lap, 2, hits.data_ptr(), cap, cnt.data_ptr(),
                                               scratch.data_ptr(), ob, hs))
    out = {}
    for name, fn in (("scan", plain), ("scan_ordered", ordered)):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name + "_us"] = round(a.elapsed_time(b) / 10 * 1e3, 1)
    n = int(cnt.item())
    tail = (cap * 4 + 255) // 256 * 256
    work = scratch[ob - tail:].view(torch.int32)[:n]
    out["records"] = n
    out["share_a_bucket"] = int((work != -1).sum().item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
