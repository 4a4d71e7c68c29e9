#!/usr/bin/env python3
"""Hop selection / CLK1-27 reversal timings quoted in DESIGN.md (run on the MI355X box):
  * whole 2^27-entry pattern generation (hop_sequence_kernel): ms, GB/s written
  * btbbx_hop_reversal_open (init_candidates) and the winnowing calls, wall clock
  * the same steps on the host CPU with the oracle (one core) for scale
Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import libbtbb_amd as bt
import _hop
import _libs


def main():
    out = {}
    bt.init(2)
    lib = bt.lib()
    hs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    lap, uap = 0x9E8B33, 0x47
    for name, amap in (("basic", None), ("afh57", _hop.afh_map_bytes(rng, 57))):
        cfg = bt.hop_cfg(lap, uap, amap)
        d = torch.empty(1 << 27, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            bt.check(lib.btbbx_hop_sequence_device(C.byref(cfg), 0, 1 << 27, d.data_ptr(), hs))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            lib.btbbx_hop_sequence_device(C.byref(cfg), 0, 1 << 27, d.data_ptr(), hs)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out["sequence_" + name] = {"ms": round(ms, 4), "GB_s_written": round((1 << 27) / ms / 1e6, 1)}
        seq = d.cpu().numpy()

        # reversal: open + winnow until one candidate is left
        c0 = int(rng.integers(0, 1 << 27))
        obs = _hop.observations(rng, seq, c0, 16)
        t0 = time.perf_counter()
        rev = bt.HopReversal(cfg, c0 & 63, obs[0][1])
        t_open = time.perf_counter() - t0
        n0 = rev.count
        steps = []
        for k in range(1, len(obs)):
            t0 = time.perf_counter()
            stop, count, cand0 = rev.winnow([obs[k][0]], [obs[k][1]])
            steps.append((count, round((time.perf_counter() - t0) * 1e6, 1)))
            if count <= 1:
                break
        assert count == 1 and cand0 == c0
        rev.close()
        # a second open: buffers are allocated per handle, so this is the steady cost too
        t0 = time.perf_counter()
        rev = bt.HopReversal(cfg, c0 & 63, obs[0][1])
        t_open2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        stop, count, cand0 = rev.winnow([o[0] for o in obs[1:]], [o[1] for o in obs[1:]])
        t_all = time.perf_counter() - t0
        assert count == 1 and cand0 == c0
        rev.close()
        out["reversal_" + name] = {"candidates": n0, "open_us": round(t_open * 1e6, 1), "open_again_us": round(t_open2 * 1e6, 1),
                                   "winnow_steps_count_us": steps, "winnow_all_at_once_us": round(t_all * 1e6, 1)}

        # host CPU, oracle (materialised table like the reference), one core
        orc = _libs.oracle()
        t0 = time.perf_counter()
        pn, oseq = _hop.orc_pattern(orc, lap, uap, amap)
        t_gen = time.perf_counter() - t0
        assert np.array_equal(oseq, seq)
        c = pn.contents
        c.first_pkt_time, c.clk_offset = 0, c0 & 63
        c.pattern_indices[0], c.pattern_channels[0] = obs[0]
        c.packets_observed = 1
        t0 = time.perf_counter()
        n_cpu = orc.orc_init_hop_reversal(0, pn)
        t_init = time.perf_counter() - t0
        assert n_cpu == n0
        cpu_steps = []
        for k in range(1, len(obs)):
            c.pattern_indices[c.packets_observed], c.pattern_channels[c.packets_observed] = obs[k]
            c.packets_observed += 1
            t0 = time.perf_counter()
            rv = orc.orc_winnow(pn)
            cpu_steps.append((rv, round((time.perf_counter() - t0) * 1e6, 1)))
            if rv <= 1:
                break
        out["cpu_oracle_" + name] = {"gen_hops_ms": round(t_gen * 1e3, 1), "init_candidates_us": round(t_init * 1e6, 1),
                                     "winnow_steps_count_us": cpu_steps}
        orc.orc_piconet_free(pn)
        orc.orc_hop_cache_clear()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
