"""The N > 1 branch of bench.py on real kernels: two torch.distributed ranks (gloo rendezvous on 127.0.0.1, both on
cuda:0 -- one GPU per gpurun box) scan their shards of one logical capture through the PRODUCT library; the union of
their hit lists must be the hit list of a single scan of the whole capture, for both layouts bench.py knows:
one stream cut in time (BASELINE configs[1] form) and 79 channels each cut in time (configs[3] form).
SURVEY.md 8(e): shards are independent, no collective on the data path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_ranks(tmp_path, extra, world=2):
    prefix = str(tmp_path / "hits")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29517 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--share-gpu", "--no-cpu", "--no-secondary", "--dump-hits", prefix] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    parts = [np.load("%s.rank%d.npy" % (prefix, k)) for k in range(world)]
    return json.loads(line), parts


def run_two_ranks(tmp_path, extra):
    return run_ranks(tmp_path, extra, 2)


def single_scan(n_streams, n_words, first_words, search_bits):
    """One launch over the same logical capture: row c = global words [first_words[c], + n_words)."""
    import torch
    import bench
    import libbtbb_amd as bt
    bt.init(2)
    lib = bt.lib()
    buf = torch.empty(n_streams * n_words, dtype=torch.int64, device="cuda")
    for c in range(n_streams):
        bt.check(lib.btbbx_synth_device(buf.data_ptr() + 8 * c * n_words, first_words[c], n_words, bench.SEED, bench.STRIDE, -1, 4, None))
    cap = n_streams * (search_bits // bench.STRIDE + 64) + (1 << 16)
    hits = torch.zeros(cap * 2, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    bt.check(lib.btbbx_scan_device(buf.data_ptr(), n_words, n_words, n_streams, search_bits, bt.LAP_ANY, 2, hits.data_ptr(), cap,
                                   cnt.data_ptr(), None))
    torch.cuda.synchronize()
    n = int(cnt.item())
    assert n <= cap
    bt.check(lib.btbbx_sort_hits_device(hits.data_ptr(), n, None))
    return hits.cpu().numpy().view(bt.HIT_DTYPE)[:n]


def merged(parts):
    allh = np.concatenate(parts)
    return allh[np.lexsort((allh["offset"], allh["stream"]))]


def test_two_ranks_time_sharded_single_stream(tmp_path):
    gib = 0.25
    line, parts = run_two_ranks(tmp_path, ["--gib", str(gib)])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    words_per_gpu = int(gib * (1 << 30)) // 8
    total_bits = 2 * words_per_gpu * 64 - 63
    assert line["config"]["symbols_per_gpu"] * 2 >= total_bits
    want = single_scan(1, 2 * words_per_gpu, [0], total_bits)
    got = merged(parts)
    assert len(parts[0]) > 10000 and len(parts[1]) > 10000
    for f in ("stream", "offset", "lap", "ac_errors"):
        assert np.array_equal(got[f], want[f]), f


def test_two_ranks_79_channels(tmp_path):
    gib = 0.5
    line, parts = run_two_ranks(tmp_path, ["--layout", "channels79", "--gib", str(gib)])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and cfg["hit_counts_match_ground_truth"]
    assert len(cfg["per_rank"]) == 2 and all(r["hits"] >= r["injected_hits"] > 1000 for r in cfg["per_rank"])
    W = int(gib * (1 << 30)) // 8 // 79
    want = single_scan(79, W, [c * W for c in range(79)], W * 64 - 63)
    got = merged(parts)
    assert cfg["hits_total"] == len(got) == len(want)
    for f in ("stream", "offset", "lap", "ac_errors"):
        assert np.array_equal(got[f], want[f]), f


def test_eight_ranks_time_sharded_single_stream(tmp_path):
    """The launch the driver makes on an 8-GPU node (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8), with
    the eight ranks on this box's one GPU: eight shards of one logical capture, every seam between two ranks, rank numbers
    above 1 in the plan, the max-over-ranks timing and the rank-0 line."""
    gib = 0.0625
    line, parts = run_ranks(tmp_path, ["--gib", str(gib)], world=8)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    words_per_gpu = int(gib * (1 << 30)) // 8
    total_bits = 8 * words_per_gpu * 64 - 63
    want = single_scan(1, 8 * words_per_gpu, [0], total_bits)
    got = merged(parts)
    assert all(len(p) > 1000 for p in parts)
    for f in ("stream", "offset", "lap", "ac_errors"):
        assert np.array_equal(got[f], want[f]), f
    # every seam is covered exactly once: the hit closest below each shard boundary and the one above it are both there
    for r in range(1, 8):
        seam = r * words_per_gpu * 64
        assert (want["offset"] < seam).any() and (want["offset"] >= seam).any()


def test_eight_ranks_79_channels(tmp_path):
    gib = 0.5
    line, parts = run_ranks(tmp_path, ["--layout", "channels79", "--gib", str(gib)], world=8)
    cfg = line["config"]
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and cfg["hit_counts_match_ground_truth"]
    assert len(cfg["per_rank"]) == 8 and sorted(r["rank"] for r in cfg["per_rank"]) == list(range(8))
    W = int(gib * (1 << 30)) // 8 // 79
    want = single_scan(79, W, [c * W for c in range(79)], W * 64 - 63)
    got = merged(parts)
    assert cfg["hits_total"] == len(got) == len(want)
    for f in ("stream", "offset", "lap", "ac_errors"):
        assert np.array_equal(got[f], want[f]), f
