"""Time / channel sharding of the scan across the GPUs of one node (host side).

The path shards with NO data-path collective: every rank owns a contiguous
slice of offsets of every stream and reads a 63-symbol halo past its end, so the union of
the per-rank hit lists is exactly the single-device hit list.  torch.distributed is used
only to gather the (sparse) hit lists / timings; nothing here touches the kernels.
"""
import numpy as np

HALO_BITS = 63            # an access code starting at the last owned offset ends 63 symbols later


def plan_py(total_search_bits, world):
    """The same arithmetic as btbbx_shard_plan (csrc/scan.hip), in Python: for hosts that plan without a GPU runtime
    (the C library needs libamdhip64 to load).  tests/test_sharding_gloo.py holds the two against each other."""
    words_total = (total_search_bits + 63) // 64
    per = (words_total + world - 1) // world
    out = []
    for r in range(world):
        w0 = min(r * per, words_total)
        w1 = min(w0 + per, words_total)
        end = min(w1 * 64, total_search_bits)
        bits = end - w0 * 64 if end > w0 * 64 else 0
        out.append({"first_word": w0, "first_offset": w0 * 64, "search_bits": bits,
                    "n_words": (bits + 63 + 63) // 64 if bits else 0})
    return out


def plan(total_search_bits, world):
    """Split offsets [0, total_search_bits) into `world` contiguous word-aligned slices.

    Returns a list of dicts {first_word, n_words, search_bits, first_offset}: rank r must hold
    words [first_word, first_word + n_words) of the stream (slice + halo) and test offsets
    [0, search_bits) of that buffer; global offset = first_offset + local offset.

    The plan itself is the C library's (btbbx_shard_plan, include/btbbx.h) -- the same one
    btbbx_scan_host_multi applies inside one process -- so the two multi-GPU forms cannot drift."""
    assert world >= 1 and total_search_bits >= 0
    from . import shard_plan
    try:
        return [shard_plan(total_search_bits, world, r) for r in range(world)]
    except OSError:                                     # the library (libamdhip64) does not load here: same plan in Python
        return plan_py(total_search_bits, world)


def merge(per_rank_hits, plans):
    """Per-rank hit arrays (HIT_DTYPE-like structured arrays with local offsets) -> one array
    with global offsets, sorted by (stream, offset)."""
    parts = []
    for hits, p in zip(per_rank_hits, plans):
        h = np.array(hits, copy=True)
        if len(h):
            h["offset"] += np.uint64(p["first_offset"])
        parts.append(h)
    allh = np.concatenate(parts) if parts else np.zeros(0)
    order = np.lexsort((allh["offset"], allh["stream"]))
    return allh[order]


def gather_hits(local_hits, dist_module=None):
    """all_gather of variable-length hit arrays over torch.distributed (any backend)."""
    import torch
    import torch.distributed as dist
    dist_module = dist_module or dist
    world = dist_module.get_world_size()
    raw = np.ascontiguousarray(local_hits).view(np.uint8).reshape(-1)
    n = torch.tensor([raw.size], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist_module.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8)
    buf[:raw.size] = torch.from_numpy(raw.copy())
    bufs = [torch.zeros(max(cap, 1), dtype=torch.uint8) for _ in range(world)]
    dist_module.all_gather(bufs, buf)
    return [b[:int(s.item())].numpy().view(local_hits.dtype) for b, s in zip(bufs, sizes)]
