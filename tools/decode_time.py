#!/usr/bin/env python3
"""decode_hits_kernel by packet type (run on the MI355X box from the repo root):

  python tools/decode_time.py [TYPES ...]     the config-3 mix, all sixteen types, and single types (e.g. DH5 DM1+DH1+DM3+FHS)

A capture like bench.py's config 3 (79 channel streams, a packet every 4096 symbols, CLK1-6 = slot number, 64 x tiled
along time: 1.29 M packets), but with the packet types and payload lengths asked for; known-LAP scan + device order
once, then btbbx_decode_hits_counted_device timed alone with HIP events.  Prints one JSON line per case."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import libbtbb_amd as bt  # noqa: E402
from libbtbb_amd import synth  # noqa: E402

MAXBODY = {synth.TYPE_DM1: 17, synth.TYPE_DH1: 27, synth.TYPE_DV: 9, synth.TYPE_AUX1: 29, synth.TYPE_DM3: 121,
           synth.TYPE_DH3: 183, synth.TYPE_DM5: 224, synth.TYPE_DH5: 339, synth.TYPE_HV1: 10, synth.TYPE_HV2: 20,
           synth.TYPE_HV3: 30, synth.TYPE_EV4: 120, synth.TYPE_EV5: 180}
NAMES = ["NULL", "POLL", "FHS", "DM1", "DH1", "HV1", "HV2", "HV3", "DV", "AUX1", "DM3", "DH3", "EV4", "EV5", "DM5", "DH5"]


def case(lib, dev, hs, types, full, seed=7):
    lap, uap = 0x9E8B33, 0x47
    nch, wpc0, tiles = 79, 1 << 14, 64
    rng = np.random.default_rng(seed)
    base = synth.noise_words(seed + 3, 0, nch * wpc0).reshape(nch, wpc0)
    slots = wpc0 * 64 // 4096 - 1
    for ch in range(nch):
        symc = synth.unpack_bits(base[ch])
        for k in range(slots):
            t_ = types[(k + ch) % len(types)]
            mb = MAXBODY.get(t_, 0)
            if full or t_ in (synth.TYPE_HV1, synth.TYPE_HV2, synth.TYPE_HV3):
                nb = mb
            elif len(types) == 1 and mb > 100:
                nb = int(rng.integers(0, mb + 1))            # a single multi-slot type, "not full": any length
            else:
                nb = min(mb, int(rng.integers(1, 17)))
            body = rng.integers(0, 256, nb, dtype=np.uint8).tobytes()
            p = synth.build_packet(lap, uap, k & 63, t_, lt_addr=1 + k % 7, flags=k % 8, body=body,
                                   voice=rng.integers(0, 256, 10, dtype=np.uint8).tobytes(),
                                   fhs_bits=synth.fhs_payload(lap, uap, 0x1234, k, rng))[:3900]
            pos = k * 4096 + 100 + int(rng.integers(0, 64))
            symc[pos:pos + len(p)] = p
        base[ch] = synth.pack_bits(symc)
    wpc = wpc0 * tiles
    d3 = torch.from_numpy(base.view(np.int64)).to(dev).repeat(1, tiles).contiguous()
    nbits = wpc * 64 - 63
    cap = nch * slots * tiles + (1 << 16)
    hits = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ln = torch.zeros(cap, dtype=torch.int32, device=dev)
    pin = torch.zeros(cap, 4, dtype=torch.int32, device=dev)
    pout = torch.zeros(cap * bt.PKTOUT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ob = lib.btbbx_order_hits_scratch_bytes(cap)
    scratch = torch.empty(ob, dtype=torch.uint8, device=dev)
    bt.check(lib.btbbx_scan_ordered_device(d3.data_ptr(), wpc, wpc, nch, nbits, lap, 2, hits.data_ptr(), cap, cnt.data_ptr(),
                                           scratch.data_ptr(), ob, hs))
    pin[:, 1] = ((hits.view(cap, 2)[:, 0] >> 12) & 63).to(torch.int32)
    pin[:, 2] = (1 << 0) | (1 << 2) | (1 << 4)
    pin[:, 3] = uap

    def dec():
        bt.check(lib.btbbx_decode_hits_counted_device(d3.data_ptr(), wpc, wpc, hits.data_ptr(), pin.data_ptr(), cnt.data_ptr(),
                                                      cap, 3125, pout.data_ptr(), ln.data_ptr(), hs))
    dec()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        dec()
    b.record()
    torch.cuda.synchronize()
    n = int(cnt.item())
    res = pout.cpu().numpy().view(bt.PKTOUT_DTYPE)[:n]
    ok = int(((res["payload_rv"] == 10) | (res["payload_rv"] == 1000) | (res["payload_rv"] == 2) | (res["payload_rv"] == 1)).sum())
    us = a.elapsed_time(b) / 5 * 1e3
    return {"types": "+".join(NAMES[t] for t in types), "payloads": "full" if full else ("0-max bytes" if len(types) == 1 and MAXBODY.get(types[0], 0) > 100 else "1-16 bytes"), "packets": n,
            "decoded": ok, "us": round(us, 1), "G_packets_s": round(n / us / 1e3, 2)}


def main():
    bt.init(2)
    lib = bt.lib()
    dev = torch.device("cuda:0")
    hs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S = synth
    cases = [([S.TYPE_DM1, S.TYPE_DH1, S.TYPE_DM3, S.TYPE_FHS], False), ([S.TYPE_DM1, S.TYPE_DH1, S.TYPE_DM3, S.TYPE_FHS], True),
             ([S.TYPE_DM1, S.TYPE_DH1, S.TYPE_DM3, S.TYPE_DH3, S.TYPE_DM5, S.TYPE_DH5, S.TYPE_FHS], True),
             (list(range(16)), False), (list(range(16)), True)]
    for t in (S.TYPE_DM1, S.TYPE_DH1, S.TYPE_FHS, S.TYPE_HV3, S.TYPE_DM3, S.TYPE_DH3, S.TYPE_DM5, S.TYPE_DH5, S.TYPE_EV4, S.TYPE_EV5):
        cases.append(([t], True))
    for t in (S.TYPE_DM3, S.TYPE_DH3, S.TYPE_DM5, S.TYPE_DH5):          # random lengths: the wave phase with mixed group sizes
        cases.append(([t], False))
    only = sys.argv[1:]
    for types, full in cases:
        if only and "+".join(NAMES[t] for t in types) not in only:
            continue
        print(json.dumps(case(lib, dev, hs, types, full)), flush=True)


if __name__ == "__main__":
    main()
