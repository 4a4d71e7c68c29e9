"""LAP_ANY over many streams in one launch (BASELINE config 4's per-GPU shard: 79 channels, 8 GiB) against one
stream of the same size: python tools/multistream_time.py (on the MI355X box)."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libbtbb_amd as bt
lib = bt.lib(); bt.init(2)
dev = torch.device("cuda:0")
def run(nst, wps):
    words = torch.empty(nst * wps, dtype=torch.int64, device=dev)
    bt.check(lib.btbbx_synth_device(words.data_ptr(), 0, nst * wps, 12345, 4096, -1, 3, None))
    cap = nst * wps * 64 // 4096 + (1 << 16)
    hits = torch.zeros(cap * 2, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    nbits = wps * 64 - 63
    def go():
        cnt.zero_()
        bt.check(lib.btbbx_scan_device(words.data_ptr(), wps, wps, nst, nbits, bt.LAP_ANY, 2, hits.data_ptr(), cap, cnt.data_ptr(), None))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(nst, wps, "ms", round(ms, 3), "Tbit/s", round(nst * nbits / ms / 1e9, 3), "hits", int(cnt.item()))
run(1, 1 << 29)
run(79, (1 << 30) // 79)          # 8 GiB over 79 streams
run(79, 1 << 20)
run(79, 100003)
