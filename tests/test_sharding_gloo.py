"""N > 1 path on CPU: two gloo processes each take one time shard (slice + 63-symbol halo)
of the same stream, scan it (here with the oracle standing in for the GPU kernel -- this
test is about the sharding/merge logic and runs without a GPU), all_gather their hit lists
and must reproduce the single-process result exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _worker(rank, world, port, total_bits, seed, out_path):
    import _libs
    import libbtbb_amd as bt
    from libbtbb_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = _libs.oracle()
    orc.orc_init(2)
    plans = shard.plan(total_bits, world)
    p = plans[rank]
    # each rank regenerates only its own slice of the logical stream (counter-based generator)
    words, _ = synth.make_stream(seed, max(p["n_words"], 1), stride=1024, first_word=p["first_word"])
    sym = np.ascontiguousarray(synth.unpack_bits(words))
    local = _libs.orc_find_all(sym, p["search_bits"], _libs.LAP_ANY, 2)
    hits = np.zeros(len(local), dtype=bt.HIT_DTYPE)
    for i, (o, l, e) in enumerate(local):
        hits[i] = (o, l, e, 0, 0)
    parts = shard.gather_hits(hits)
    merged = shard.merge(parts, plans)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world
    if rank == 0:
        np.save(out_path, merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_words", [4096 + 7, 1000])
def test_two_rank_sharding_matches_single_scan(tmp_path, total_words):
    import _libs
    import libbtbb_amd as bt
    from libbtbb_amd import shard, synth
    seed, world = 77, 2
    total_bits = total_words * 64 - 63 - 11          # not word aligned on purpose
    out = str(tmp_path / "merged.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, total_bits, seed, out), nprocs=world, join=True,
                       start_method="spawn")
    merged = np.load(out)
    words, _ = synth.make_stream(seed, total_words, stride=1024)
    sym = np.ascontiguousarray(synth.unpack_bits(words))
    orc = _libs.oracle()
    orc.orc_init(2)
    want = _libs.orc_find_all(sym, total_bits, _libs.LAP_ANY, 2)
    got = [(int(h["offset"]), int(h["lap"]), int(h["ac_errors"])) for h in merged]
    assert got == want and len(got) > 30


def test_plan_covers_every_offset_once():
    from libbtbb_amd import shard
    for total in (0, 1, 63, 64, 65, 1000, 64 * 1000 + 5, 1 << 35):
        for world in (1, 2, 3, 4, 8):
            plans = shard.plan(total, world)
            assert plans == shard.plan_py(total, world)          # C plan (btbbx_shard_plan) == the Python arithmetic
            assert len(plans) == world
            pos = 0
            for p in plans:
                if p["search_bits"]:
                    assert p["first_offset"] == pos
                    assert p["search_bits"] + 63 <= p["n_words"] * 64
                    pos += p["search_bits"]
            assert pos == total
