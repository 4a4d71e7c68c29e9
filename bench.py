#!/usr/bin/env python3
"""bench.py -- Gbit/s of raw bitstream scanned (LAP_ANY, max_ac_errors = 2) on MI355X.

Workload (BASELINE.json configs[1]): promiscuous LAP_ANY access-code scan of a 4 GiB
single-channel packed bitstream resident in HBM (2^35 symbols; iid noise plus one ID-packet
sync word with 0..3 bit errors every 4096 symbols, generated on the device by
btbbx_synth_device).  One "step" = one pass of the scan kernel over the whole shard.
With --gpus N the logical capture is N x 4 GiB and rank r takes shard r of the library's own
time-shard plan (btbbx_shard_plan: word-aligned slice + 63-symbol halo, the plan
btbbx_scan_host_multi applies inside one process) -- weak scaling, no data-path collective;
torch.distributed is used for the barrier and for the max-over-ranks of the timing only.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(nbits / 8 read + 16 B per reported hit written) / mean kernel time from HIP events on the
launch stream, against the 8 TB/s HBM peak.  `cpu_baseline` = the UNMODIFIED reference
(oracle/_ref/libbtbb_ref.so, all-matches loop around btbb_find_ac) -- or, if that file is
missing, the oracle port -- timed on ALL of this box's host threads over a bounded slice of
the same stream; its hit list is also compared with the GPU's (field "parity").

`secondary` (N = 1 only, timed in the same run, after the headline region): BASELINE configs 3 and
5 -- the known-LAP full chain over 79 hop-channel streams (scan -> sort -> gather -> header +
payload decode) and the 64-seed CLK1-6 brute force over a stream of 2^20 detected packets -- each
with its own roofline figure and the reference's CPU loop on a bounded sample beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_MIN_PER_WORD = 124        # fewest vector instructions per 64 offsets of an exact LAP_ANY filter of this shape (NOTEBOOK.md 6.4 for the derivation: 30 + 30 + 8 x 8)
SEED = 20260926
STRIDE = 4096


def csrc_fingerprint(csrc_dir=None):
    """sha256 (16 hex digits) over the kernel / library sources the loaded .so is built from (libbtbb_amd/csrc: *.hip,
    *.cpp, *.h, Makefile, in name order) with comments and white space taken out, so that only changes the compiler
    sees change it.  profiles/traffic*.json carry the fingerprint of the build their PMC passes measured; a line whose
    build differs prints traffic: null instead of a number that belongs to another kernel."""
    import hashlib
    import re
    d = csrc_dir or os.path.join(ROOT, "libbtbb_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")) or name == "Makefile":
            text = open(os.path.join(d, name), "r", errors="replace").read()
            if name != "Makefile":
                text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
                text = re.sub(r"//[^\n]*", " ", text)                       # line comments
            else:
                text = re.sub(r"#[^\n]*", " ", text)
            h.update(name.encode() + b"\0" + " ".join(text.split()).encode())
    return h.hexdigest()[:16]


def host_cpu():
    """What the CPU baseline runs on: model string, logical threads, physical cores."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    return {"model": model, "threads": threads, "physical_cores": len(cores) or None}


def physical_cpus():
    """One logical CPU per physical core (the first SMT sibling of each), restricted to this process's affinity."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    seen, firsts = set(), []
    for cpu in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            firsts.append(cpu)
    return allowed, firsts


def pin_to_gpu_numa(device_index):
    """One process per GPU: keep this rank's host threads (launches, the event loop of torch.distributed) on the NUMA node
    the GPU hangs off -- /sys/bus/pci/devices/<bus id>/numa_node, the node's cpulist cut with what the process may use.
    Returns what was done, for the JSON line; never fails the run."""
    info = {"numa_node": None, "cpus": None}
    try:
        props = torch.cuda.get_device_properties(device_index)
        bus = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        info["numa_node"] = node
        if node < 0:
            return info
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if allowed:
            os.sched_setaffinity(0, allowed)
            info["cpus"] = len(allowed)
    except Exception as e:                      # no sysfs entry, old torch, a container without the node files ...
        info["error"] = str(e)[:80]
    return info


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(words_host, first_word, gpu_hits, cpu):
    """The UNMODIFIED reference's all-matches scan (oracle/_ref, refint_find_all_mt: pthreads behind a barrier, each
    worker timed with CLOCK_MONOTONIC around its native loop only -- no interpreter, allocation or unpacking inside
    the timed region) over `words_host`: on every logical CPU, on one thread per physical core, and on one thread
    alone.  Falls back to the oracle port (Python thread pool) when the compiled reference is not there."""
    import _libs
    from libbtbb_amd import synth
    ref = _libs.ref()
    lo_bit = first_word * 64
    if ref is None:                                    # port: liboracle.so, one slice per thread
        orc = _libs.oracle()
        orc.orc_init(2)
        cores = cpu["threads"]
        sym = np.ascontiguousarray(synth.unpack_bits(words_host))
        n = len(sym) - 63
        bounds = np.linspace(0, n, cores + 1).astype(np.int64)

        def work(i):
            lo, hi = int(bounds[i]), int(bounds[i + 1])
            return [(o + lo, l, e) for (o, l, e) in _libs.orc_find_all(np.ascontiguousarray(sym[lo:hi + 63]), hi - lo, 0xFFFFFFFF, 2)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            parts = list(ex.map(work, range(cores)))
        dt = time.perf_counter() - t0
        cpu_hits = [h for p in parts for h in p]
        sel = gpu_hits[(gpu_hits["offset"] >= lo_bit) & (gpu_hits["offset"] < lo_bit + n)]
        sel = sel[np.argsort(sel["offset"], kind="stable")]
        gpu_list = [(int(h["offset"]) - lo_bit, int(h["lap"]), int(h["ac_errors"])) for h in sel]
        rate = n / dt / 1e9
        return {"value": round(rate, 4), "unit": "Gbit/s", "cores": cores, "kind": "port", "cpu_model": cpu["model"],
                "physical_cores": cpu["physical_cores"], "per_thread_Msym_s": round(rate * 1e3 / cores, 2),
                "sample": "first %d symbols, %d Python threads around liboracle.so (timing includes the thread pool)" % (n, cores),
                "hits": len(cpu_hits)}, cpu_hits == gpu_list

    ref.btbb_init(2)
    allowed, firsts = physical_cpus()
    t0 = time.perf_counter()
    sym = _libs.ref_unpack_mt(words_host, min(64, len(allowed)))       # one symbol per byte (the reference's layout)
    unpack_s = time.perf_counter() - t0
    n = len(sym) - 63

    def run(cpus, pinned):
        off, laps, errs, found, secs, wall = _libs.ref_find_all_mt(sym, n, 0xFFFFFFFF, 2, len(cpus), cpus if pinned else None)
        per = (n / len(cpus)) / secs / 1e6                                # Msym/s of each worker over its own slice
        return {"threads": len(cpus), "pinned": bool(pinned), "Gbit_s": round(n / wall / 1e9, 4), "wall_s": round(wall, 4),
                "per_thread_Msym_s": {"min": round(float(per.min()), 2), "mean": round(float(per.mean()), 2),
                                      "max": round(float(per.max()), 2)},
                "thread_seconds": {"min": round(float(secs.min()), 4), "max": round(float(secs.max()), 4)}}, (off, laps, errs)

    every, hits_every = run(allowed, pinned=True)                      # one worker per logical CPU
    phys, _ = run(firsts, pinned=True) if len(firsts) < len(allowed) else (None, None)
    # A container with a CPU quota (cgroup cpu.max; the gpurun boxes grant 16 CPUs of a 256-thread host) gets no more
    # out of more runnable threads -- the scheduler throttles them (tools/cpu_scaling.py: linear up to the quota,
    # flat or falling beyond).  The honest "all the host cores we may use" figure is one worker per granted CPU.
    quota = cpu_quota()
    granted = None
    if quota is not None and int(quota) >= 1 and int(quota) < len(firsts):
        granted, _ = run(firsts[:int(quota)], pinned=True)
    n_solo = min(n, 1 << 26)                                           # one undisturbed thread: ~0.5-1 s
    off1, _, _, _, secs1, _ = _libs.ref_find_all_mt(sym, n_solo, 0xFFFFFFFF, 2, 1, [firsts[0]])
    solo = n_solo / float(secs1[0]) / 1e6

    best = max([r for r in (every, phys, granted) if r is not None], key=lambda r: r["Gbit_s"])
    sel = gpu_hits[(gpu_hits["offset"] >= lo_bit) & (gpu_hits["offset"] < lo_bit + n)]
    sel = sel[np.argsort(sel["offset"], kind="stable")]
    off, laps, errs = hits_every
    parity = (len(sel) == len(off) and bool(np.array_equal(sel["offset"].astype(np.uint64) - np.uint64(lo_bit), off))
              and bool(np.array_equal(sel["lap"].astype(np.uint32), laps))
              and bool(np.array_equal(sel["ac_errors"].astype(np.uint8), errs)))
    # "cores" = the CPUs this container may use: what its cgroup grants when it has a quota (16 of the host's 256 threads on the
    # gpurun boxes), else the logical CPUs it is allowed on; "threads" = the workers of the placement whose rate is quoted
    # (more runnable threads than granted CPUs are throttled, not added)
    cores_granted = int(quota) if quota is not None and 1 <= int(quota) < len(allowed) else len(allowed)
    return {
        "value": best["Gbit_s"], "unit": "Gbit/s", "cores": cores_granted, "threads": best["threads"], "kind": "reference",
        "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"], "host_logical_cpus": len(allowed),
        "per_thread_Msym_s": best["per_thread_Msym_s"]["mean"], "solo_thread_Msym_s": round(solo, 2),
        "all_logical_cpus": every, "one_thread_per_core": phys, "one_thread_per_granted_cpu": granted,
        "cgroup_cpu_quota": quota,
        "sample": "first %d symbols of the same stream (one symbol per byte; native unpack %.2f s excluded), the reference's "
                  "btbb_find_ac in the all-matches loop on pinned pthreads started behind a barrier, CLOCK_MONOTONIC around "
                  "each thread's native loop only; value = the fastest of {every logical CPU, one thread per physical core, one thread "
                  "per CPU of the container's cgroup quota}; "
                  "solo = one thread alone over %d symbols" % (n, unpack_s, n_solo),
        "hits": int(len(off)),
    }, parity


# one record per access code from oracle/ref_internals.c refint_known_lap_chain_records (the reference's chain, natively looped)
CHAIN_REC = np.dtype([("offset", "<u8"), ("payload_hash", "<u8"), ("payload_rv", "<i4"), ("payload_length", "<i4"), ("ac_errors", "u1"),
                      ("header_rv", "u1"), ("type", "u1"), ("lt_addr", "u1"), ("hdr_flags", "u1"), ("hec", "u1"),
                      ("header_present", "u1"), ("pad", "u1")])


def payload_hash(res):
    """refint_known_lap_chain_records' hash of the payload bits a decoder left, from btbbx_pkt_out records: the
    payload_length * 8 bits as LSB-first words w_k, sum of w_k * (2 k + 1) mod 2^64; 0 unless the payload decoder
    returned 2 / 10 / 1000."""
    nb = res["payload_length"].astype(np.int64) * 8
    k = np.arange(43, dtype=np.int64)
    inw = np.clip(nb[:, None] - 64 * k[None, :], 0, 64)
    mask = np.where(inw >= 64, np.uint64(0xFFFFFFFFFFFFFFFF),
                    (np.uint64(1) << np.minimum(inw, 63).astype(np.uint64)) - np.uint64(1))
    h = ((res["payload"] & mask) * (2 * k + 1).astype(np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
    wrote = (res["header_rv"] == 1) & np.isin(res["payload_rv"], (2, 10, 1000))
    return np.where(wrote, h, np.uint64(0))


class Timer:
    """HIP-event timing on the stream the kernels are launched on (torch's current stream, whose
    handle is what the C ABI receives)."""

    def __init__(self, stream):
        self.s = stream

    def ms(self, fn, reps, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(self.s)
        for _ in range(reps):
            fn()
        b.record(self.s)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps


_TSEC = None


def secondary_traffic(name):
    """(bytes per step, source, VALU block) of one secondary line from profiles/traffic_secondary.json -- the last rocprofv3 --pmc
    passes over `bench.py --only-secondary <line>` (tools/collect_evidence.sh), NOT measured in this run; null when the build
    this run loads differs from the sources that were measured."""
    global _TSEC
    if _TSEC is None:
        try:
            _TSEC = json.load(open(os.path.join(ROOT, "profiles", "traffic_secondary.json")))
        except (OSError, ValueError):
            _TSEC = {}
    e = _TSEC.get(name)
    if not e:
        return None, None, None
    if _TSEC.get("csrc_sha16") != csrc_fingerprint():
        return None, "null: profiles/traffic_secondary.json was measured on sources %s, this build is %s" % (
            _TSEC.get("csrc_sha16"), csrc_fingerprint()), None
    return (int(e["bytes_per_step"]), "profiles/traffic_secondary.json (%s); not re-measured in this run" % _TSEC.get("source", "rocprofv3 --pmc"),
            e.get("valu"))


def valu_block(v, kernel_ms):
    """roofline.valu of a line: the vector-issue time of its dominant kernel = wave-instructions per launch (PMC run) x cycles per
    instruction (the kernel's static mix priced with tools/valu_rate.hip) / (SIMDs x clock), as a fraction of the kernel's time."""
    if not v or not v.get("insts_per_launch"):
        return None
    busy_s = v["insts_per_launch"] * v["cycles_per_inst"] / (v["simds"] * v["clock_ghz"] * 1e9)
    return {"kernel": v.get("kernel"), "insts_per_launch": int(v["insts_per_launch"]), "cycles_per_inst": v["cycles_per_inst"],
            "issue_ms": round(busy_s * 1e3, 4), "frac": round(busy_s / (kernel_ms * 1e-3), 4),
            "salu_per_valu": round((v.get("salu_insts_per_launch") or 0) / v["insts_per_launch"], 3),
            "wait_any_frac": v.get("wait_any_frac"),
            # measured in the same PMC run (tools/pmc_pipe.sh): SQ_THREAD_CYCLES_VALU against SQ_CYCLES -- the share of the SIMDs' cycles
            # the vector pipe was busy, and its cycles per vector instruction
            "pipe_busy_frac": v.get("pipe_busy_frac"), "pipe_cycles_per_inst": v.get("pipe_cycles_per_inst")}


def secondary(bt, lib, dev, cur, hs, cpu, with_cpu, only=None):
    """BASELINE configs 3 and 5, driver-timed: see the module docstring.  only: run just the named line (and what it needs
    as input) -- for the PMC passes of tools/collect_evidence.sh, whose per-kernel means must not mix two workloads."""
    import _libs
    from libbtbb_amd import synth
    tm = Timer(cur)
    out = {}
    traffic_of = secondary_traffic
    ref = _libs.ref() if with_cpu else None
    if ref is not None:
        ref.btbb_init(2)

    lap, uap = 0x9E8B33, 0x47
    nch, wpc0, tiles = 79, 1 << 14, 64
    slots = wpc0 * 64 // 4096 - 1
    wpc = wpc0 * tiles
    nbits = wpc * 64 - 63
    cap = nch * slots * tiles + (1 << 16)
    assert wpc * 64 < 2 ** 31
    hits = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    pk = torch.zeros(cap * 50, dtype=torch.int64, device=dev)
    ln = torch.zeros(cap, dtype=torch.int32, device=dev)
    pout = torch.zeros(cap * bt.PKTOUT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    order_bytes = lib.btbbx_scan_ordered_scratch_bytes(nbits, nch, lap, cap)     # (with the segment slots: 16 B per 4096 offsets)
    order_scratch = torch.empty(order_bytes, dtype=torch.uint8, device=dev)
    # what every packet of the piconet enters the decoders with: WHITENED | UAP_VALID | CLK6_VALID, the piconet's UAP; the
    # clock of a packet = the slot number of its access code (offset / 4096: the capture's rule), worked out by the
    # decode kernel itself (btbbx_decode_hits_piconet_device) -- round 3 filled a btbbx_pkt_in per packet with a torch kernel
    entry_in = np.zeros(1, bt.PKTIN_DTYPE)
    entry_in["flags"], entry_in["uap"] = (1 << 0) | (1 << 2) | (1 << 4), uap
    entry_ptr = entry_in.ctypes.data_as(C.c_void_p)
    ncores = cpu["threads"]
    quota = cpu_quota()
    if quota is not None and 1 <= int(quota) < ncores:
        ncores = int(quota)                                   # the CPUs the container's cgroup grants (16 of 256 on the gpurun boxes)
    MAXBODY = {synth.TYPE_DM1: 17, synth.TYPE_DH1: 27, synth.TYPE_DM3: 121, synth.TYPE_DH3: 183, synth.TYPE_DM5: 224, synth.TYPE_DH5: 339}
    NAMES = {synth.TYPE_DM1: "DM1", synth.TYPE_DH1: "DH1", synth.TYPE_DM3: "DM3", synth.TYPE_DH3: "DH3", synth.TYPE_DM5: "DM5",
             synth.TYPE_DH5: "DH5", synth.TYPE_FHS: "FHS"}
    def known_lap_chain(name, types, full, seed):
        """One config-3 capture: 79 channel streams, 2^14 words per channel built on the host (real packets of one piconet
        every 4096 symbols, CLK1-6 = slot number mod 64, whitened) and tiled 64 x along time on the device; the chain
        scan (ordered) -> decode straight from the streams, queued on one stream, timed over `reps` steps."""
        rng = np.random.default_rng(seed)
        base = synth.noise_words(seed + 3, 0, nch * wpc0).reshape(nch, wpc0)
        t_build = time.perf_counter()
        for ch in range(nch):
            symc = synth.unpack_bits(base[ch])
            for k in range(slots):
                t_ = types[(k + ch) % len(types)]
                nb = MAXBODY.get(t_, 0) if full else int(rng.integers(1, 17))
                body = rng.integers(0, 256, nb, dtype=np.uint8).tobytes()
                p = synth.build_packet(lap, uap, k & 63, t_, lt_addr=1 + k % 7, flags=k % 8, body=body,
                                       fhs_bits=synth.fhs_payload(lap, uap, 0x1234, k, rng))
                pos = k * 4096 + 100 + int(rng.integers(0, 64))
                symc[pos:pos + len(p)] = p
            base[ch] = synth.pack_bits(symc)
        t_build = time.perf_counter() - t_build
        d3 = torch.from_numpy(base.view(np.int64)).to(dev).repeat(1, tiles).contiguous()      # (79, wpc)

        def scan():
            cnt.zero_()
            bt.check(lib.btbbx_scan_device(d3.data_ptr(), wpc, wpc, nch, nbits, lap, 2, hits.data_ptr(), cap, cnt.data_ptr(), hs))

        def chain():
            # scan -> (stream, offset) order -> decode, all queued on one stream: the number of hits never leaves the
            # device (the ordering and the decode read the scan's counter from HBM)
            cnt.zero_()
            bt.check(lib.btbbx_scan_ordered_device(d3.data_ptr(), wpc, wpc, nch, nbits, lap, 2, hits.data_ptr(), cap, cnt.data_ptr(),
                                                   order_scratch.data_ptr(), order_bytes, hs))
            # header + payload decode straight from the streams (no 400-byte row per packet in between)
            bt.check(lib.btbbx_decode_hits_piconet_device(d3.data_ptr(), wpc, wpc, hits.data_ptr(), cnt.data_ptr(), cap, entry_ptr,
                                                          4096, 3125, pout.data_ptr(), ln.data_ptr(), hs))

        def decode_only():
            bt.check(lib.btbbx_decode_hits_piconet_device(d3.data_ptr(), wpc, wpc, hits.data_ptr(), cnt.data_ptr(), cap, entry_ptr,
                                                          4096, 3125, pout.data_ptr(), ln.data_ptr(), hs))
        pout.zero_()
        chain()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            chain()
        torch.cuda.synchronize()
        chain_ms = (time.perf_counter() - t0) / reps * 1e3
        n3 = int(cnt.item())                                    # read once, after the timed region
        assert n3 <= cap
        hh = hits.cpu().numpy().view(bt.HIT_DTYPE)[:n3].copy()
        k_sorted = (hh["stream"].astype(np.uint64) << np.uint64(48)) | hh["offset"]
        assert bool(np.all(k_sorted[1:] > k_sorted[:-1])), "device order is not strictly increasing in (stream, offset)"
        decode_ms = tm.ms(decode_only, 5)
        res = pout.cpu().numpy().view(bt.PKTOUT_DTYPE)[:n3].copy()
        scan_ms = tm.ms(scan, 5)                                # (leaves an unordered list behind: taken last)
        ok = (res["payload_rv"] == 10) | (res["payload_rv"] == 1000)
        pay_bytes = float(res["payload_length"][ok].sum())
        alg = nch * nbits / 8 + 16 * n3 + n3 * (391 + 32) + pay_bytes
        scan_alg = nch * nbits / 8 + 16 * n3
        entry = {
            "config": "BASELINE configs[2]: known-LAP full decode chain (find_ac -> order on the device -> header + payload "
                      "decode from the streams) over 79 hop-channel streams, %.2f GiB packed in HBM, one GPU; packets %s, %s"
                      % (nch * wpc * 8 / 2**30, " / ".join(NAMES[t] for t in types),
                         "every payload at its type's full length" if full else "bodies of 1 .. 16 bytes"),
            "value": round(nch * nbits / (chain_ms * 1e-3) / 1e9, 1), "unit": "Gbit/s", "ms_per_step": round(chain_ms, 3),
            "packets": n3, "packets_per_s": round(n3 / (chain_ms * 1e-3)), "crc_ok": int(ok.sum()),
            "payload_bytes_decoded": int(pay_bytes),
            "roofline": {"bound": "hbm", "achieved": round(alg / (chain_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(alg / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes_per_step": int(alg),
                         "kernel": "scan_known_lap_kernel", "kernel_ms": round(scan_ms, 4),
                         "kernel_frac": round(scan_alg / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "decode_hits_kernel_ms": round(decode_ms, 4),
                         "decode_hits_kernel_frac": round((n3 * (391 + 32 + 40) + 2 * pay_bytes) / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": traffic_of(name)[0], "traffic_source": traffic_of(name)[1],
                         "valu": valu_block(traffic_of(name)[2], scan_ms)},
            "host_build_s": round(t_build, 2),
        }
        if ref is not None:
            syms = [np.ascontiguousarray(synth.unpack_bits(base[ch])) for ch in range(nch)]
            recs = [np.zeros(slots + 64, CHAIN_REC) for _ in range(nch)]

            def one(ch):
                return int(ref.refint_known_lap_chain_records(_libs.ptr(syms[ch]), len(syms[ch]), lap, 2, uap, 4096,
                                                              recs[ch].ctypes.data_as(C.c_void_p), len(recs[ch])))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=ncores) as ex:
                counts = list(ex.map(one, range(nch)))
            dt = time.perf_counter() - t0
            assert all(c <= slots + 64 for c in counts)
            want = np.concatenate([recs[ch][:counts[ch]] for ch in range(nch)])
            want_stream = np.concatenate([np.full(counts[ch], ch, np.uint16) for ch in range(nch)])
            # the same channels on the GPU: tile 0 = offsets below wpc0 * 64 - 63 of every stream, in the device's order
            first = hh["offset"] < wpc0 * 64 - 63
            g, gh = res[first], hh[first]
            same = len(g) == len(want)
            if same:
                hdr = want["header_rv"] == 1
                same = bool(np.array_equal(gh["stream"], want_stream) and np.array_equal(gh["offset"], want["offset"])
                            and np.array_equal(gh["ac_errors"], want["ac_errors"])
                            and np.array_equal(g["header_present"], want["header_present"])
                            and np.array_equal(g["header_rv"], want["header_rv"].astype(np.int32))
                            and np.array_equal(g["payload_rv"], want["payload_rv"])
                            and all(np.array_equal(g[f][hdr], want[f][hdr]) for f in ("payload_length", "type", "lt_addr", "hdr_flags", "hec"))
                            and np.array_equal(payload_hash(g), want["payload_hash"]))
            entry["cpu_baseline"] = {
                "value": round(nch * (wpc0 * 64 - 63) / dt / 1e9, 4), "unit": "Gbit/s", "cores": ncores,
                "kind": "reference", "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"],
                "packets_per_s": round(sum(counts) / dt),
                "sample": "the untiled capture (79 channels x 2^20 symbols): btbb_find_ac all-matches loop + "
                          "btbb_packet_set_data + btbb_header_present + btbb_decode_header + btbb_decode_payload per match, "
                          "one channel per task on %d threads (the CPUs the container's cgroup grants)" % ncores,
            }
            entry["parity"] = same
            entry["parity_checked"] = ("%d packets of tile 0, per packet: stream, offset, ac_errors, header_present, header_rv, payload_rv, "
                                       "payload_length, type, lt_addr, flags, hec and a hash of every payload bit against the reference's"
                                       % len(want))
        return entry, d3, n3

    t4 = [synth.TYPE_DM1, synth.TYPE_DH1, synth.TYPE_DM3, synth.TYPE_FHS]
    t7 = [synth.TYPE_DM1, synth.TYPE_DH1, synth.TYPE_DM3, synth.TYPE_DH3, synth.TYPE_DM5, synth.TYPE_DH5, synth.TYPE_FHS]
    want = lambda name: only is None or only == name                  # noqa: E731
    if want("known_lap_79ch_chain_full_payloads"):
        out["known_lap_79ch_chain_full_payloads"], d3, n3 = known_lap_chain("known_lap_79ch_chain_full_payloads", t7, True, SEED + 11)
        del d3
        if only is not None:
            return out
    out["known_lap_79ch_chain"], d3, n3 = known_lap_chain("known_lap_79ch_chain", t4, False, SEED)
    if only == "known_lap_79ch_chain":
        return out
    # the packets of that capture as rows, for the config 5 stream below (the list was left unordered by the scan timing)
    bt.check(lib.btbbx_order_scan_hits_device(hits.data_ptr(), cnt.data_ptr(), cap, nch, nbits, order_scratch.data_ptr(), order_bytes, hs))
    bt.check(lib.btbbx_gather_packets_device(d3.data_ptr(), wpc, wpc, hits.data_ptr(), n3, 3125, pk.data_ptr(), ln.data_ptr(), hs))
    torch.cuda.synchronize()

    # ---- config 5: CLK1-6 / UAP brute force over a stream of 2^20 detected packets (the packets the
    # chain above gathered, tiled): 64 x {try_clock, crc_check} per packet, and the HEC-only table
    n_src = min(n3, nch * slots)
    npk = 1 << 20
    idx = (torch.arange(npk, device=dev) % n_src)
    pk5 = pk[: n3 * 50].view(n3, 50)[idx].contiguous()
    in5 = torch.zeros(npk, 4, dtype=torch.int32, device=dev)
    in5[:, 0] = ln[:n3][idx]
    in5[:, 2] = 1                                             # WHITENED; neither UAP nor clock known
    tr5 = torch.zeros(npk * 64, dtype=torch.int32, device=dev)
    tab = torch.zeros(npk * 32, dtype=torch.int32, device=dev)

    def trials():
        bt.check(lib.btbbx_trials_device(pk5.data_ptr(), in5.data_ptr(), npk, tr5.data_ptr(), hs))

    def uaptab():
        bt.check(lib.btbbx_uap_table_device(pk5.data_ptr(), in5.data_ptr(), npk, tab.data_ptr(), hs))
    if only == "clk6_bruteforce_all_types":
        out.pop("known_lap_79ch_chain", None)
    t_tr = tm.ms(trials, 3) if want("clk6_bruteforce") else 1.0
    t_u = tm.ms(uaptab, 5) if want("clk6_bruteforce") else 1.0
    alg5 = npk * (391 + 256)
    entry = {
        "config": "BASELINE configs[4]: UAP / CLK1-6 brute force, 64 whitening seeds x (HEC -> UAP, CRC check) "
                  "per packet over a stream of 2^20 detected packets (DM1 / DH1 / DM3 / FHS as captured, 400 B "
                  "packed each), one GPU",
        "value": round(npk / (t_tr * 1e-3)), "unit": "packets/s", "ms_per_step": round(t_tr, 3),
        "trials_per_s": round(npk * 64 / (t_tr * 1e-3)),
        "roofline": {"bound": "hbm", "achieved": round(alg5 / (t_tr * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(alg5 / (t_tr * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes_per_step": alg5, "kernel": "trials_linear_kernel", "kernel_ms": round(t_tr, 4),
                     "traffic": traffic_of("clk6_bruteforce")[0], "traffic_source": traffic_of("clk6_bruteforce")[1],
                     "valu": valu_block(traffic_of("clk6_bruteforce")[2], t_tr)},
        "hec_only_table": {"value": round(npk / (t_u * 1e-3)), "unit": "packets/s", "ms_per_step": round(t_u, 4),
                           "kernel": "uap_table_kernel",
                           "frac": round(npk * (8 + 128) / (t_u * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
    }
    if ref is not None:
        # bounded sample: the first 256 full-length packets of the stream as 3200-symbol buffers, one host
        # thread; the GPU side of the comparison is the 2^20-packet run above (trials_linear_kernel)
        lens_all = ln[:n_src].cpu().numpy()
        pick = np.nonzero(lens_all == 3125)[0][:256]
        m = len(pick)
        words = pk[: n_src * 50].view(n_src, 50)[torch.from_numpy(pick).to(dev)].cpu().numpy().view(np.uint64)
        buf = np.zeros((m, 3200), dtype=np.uint8)
        for i in range(m):
            buf[i] = synth.unpack_bits(words[i])
        table = np.zeros(m * 64, dtype=np.uint32)
        t0 = time.perf_counter()
        ref.refint_clk6_trials(_libs.ptr(buf), m, 3200, 3125, lap, _libs.ptr(table))
        dt = time.perf_counter() - t0
        g = tr5.view(npk, 64)[torch.from_numpy(pick).to(dev)].cpu().numpy().view(bt.TRIAL_DTYPE).reshape(-1)
        gpu_tab = g["uap"].astype(np.uint32) | (g["rv"].astype(np.int32).astype(np.uint32) << 8)
        entry["cpu_baseline"] = {
            "value": round(m / dt, 1), "unit": "packets/s", "cores": 1, "kind": "reference",
            "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"],
            "sample": "%d of the same packets, the 64-candidate try_clock + crc_check loop of btbb_uap_from_header "
                      "(bluetooth_piconet.c:675-690) on one thread" % m,
        }
        entry["parity"] = bool(np.array_equal(gpu_tab, table))
    if want("clk6_bruteforce"):
        if only is not None:
            out.pop("known_lap_79ch_chain", None)
        out["clk6_bruteforce"] = entry
        if only is not None:
            return out

    # ---- config 5 on the reference's WORST-CASE input: packets of 3125 random symbols whose header
    # region is a clean FEC-1/3 encoding of 18 random bits, so that for every candidate clock the type field is uniform
    # over 0..15 -- 89 % of the reference's time on this input goes into EV5's quadratic CRC scan (bluetooth_packet.c:1114-1126)
    n_dist = 4096
    rng5 = np.random.default_rng(SEED + 5)
    w5 = rng5.integers(0, 1 << 63, (n_dist, 50), dtype=np.int64).view(np.uint64) * np.uint64(2) + \
        rng5.integers(0, 2, (n_dist, 50), dtype=np.int64).view(np.uint64)
    hdr18 = rng5.integers(0, 1 << 18, n_dist, dtype=np.int64).astype(np.uint64)
    enc = np.zeros(n_dist, dtype=np.uint64)
    for b in range(18):                                       # FEC 1/3: every header bit three times, symbols 68 .. 121
        bit = (hdr18 >> np.uint64(b)) & np.uint64(1)
        enc |= (bit * np.uint64(7)) << np.uint64(3 * b)
    w5[:, 1] = (w5[:, 1] & ~(np.uint64((1 << 54) - 1) << np.uint64(4))) | (enc << np.uint64(4))
    w5[:, 48] &= np.uint64((1 << (3125 - 48 * 64)) - 1)       # 3125 symbols captured, zeros behind them
    w5[:, 49] = 0
    pk6 = torch.from_numpy(w5.view(np.int64)).to(dev)[torch.arange(npk, device=dev) % n_dist].contiguous()
    in6 = torch.zeros(npk, 4, dtype=torch.int32, device=dev)
    in6[:, 0] = 3125
    in6[:, 2] = 1
    del pk5

    def trials_all():
        bt.check(lib.btbbx_trials_device(pk6.data_ptr(), in6.data_ptr(), npk, tr5.data_ptr(), hs))
    t_all = tm.ms(trials_all, 3)
    entry = {
        "config": "BASELINE configs[4] on the reference's worst-case input (SURVEY.md Appendix A): 2^20 packets of 3125 random symbols "
                  "with a clean FEC-1/3 header of 18 random bits (%d distinct, tiled) -- every candidate clock sees a uniform packet "
                  "type; 64 x {try_clock, crc_check} per packet" % n_dist,
        "value": round(npk / (t_all * 1e-3)), "unit": "packets/s", "ms_per_step": round(t_all, 3),
        "trials_per_s": round(npk * 64 / (t_all * 1e-3)),
        "roofline": {"bound": "hbm", "achieved": round(alg5 / (t_all * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg5 / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": alg5,
                     "kernel": "trials_linear_kernel", "kernel_ms": round(t_all, 4),
                     "traffic": traffic_of("clk6_bruteforce_all_types")[0], "traffic_source": traffic_of("clk6_bruteforce_all_types")[1],
                     "valu": valu_block(traffic_of("clk6_bruteforce_all_types")[2], t_all)},
    }
    if ref is not None:
        m = 128
        buf = np.zeros((m, 3200), dtype=np.uint8)
        for i in range(m):
            buf[i] = synth.unpack_bits(w5[i])
        table = np.zeros(m * 64, dtype=np.uint32)
        t0 = time.perf_counter()
        ref.refint_clk6_trials(_libs.ptr(buf), m, 3200, 3125, lap, _libs.ptr(table))
        dt = time.perf_counter() - t0
        g = tr5.view(npk, 64)[:m].cpu().numpy().view(bt.TRIAL_DTYPE).reshape(-1)
        gpu_tab = g["uap"].astype(np.uint32) | (g["rv"].astype(np.int32).astype(np.uint32) << 8)
        rv_hist = {int(k): int(v) for k, v in zip(*np.unique(g["rv"], return_counts=True))}
        entry["cpu_baseline"] = {
            "value": round(m / dt, 1), "unit": "packets/s", "cores": 1, "kind": "reference", "cpu_model": cpu["model"],
            "physical_cores": cpu["physical_cores"],
            "sample": "the first %d of the same packets, the 64-candidate try_clock + crc_check loop of btbb_uap_from_header "
                      "(bluetooth_piconet.c:675-690) on one thread" % m,
        }
        entry["crc_check_rv_histogram_of_the_sample"] = rv_hist
        entry["parity"] = bool(np.array_equal(gpu_tab, table))
    out["clk6_bruteforce_all_types"] = entry
    return out


def channels79(args, bt, lib, shard, dev, rank, world, red_dev):
    """BASELINE configs[3] as written: promiscuous LAP_ANY over 79 channel streams, --gib GiB in total, every channel
    time-sharded over the `world` ranks by the library's own plan (btbbx_shard_plan: slice + 63-symbol halo, no
    collective).  The logical capture is ONE synthetic stream cut into 79 equal channels (channel c = global words
    [c W, (c + 1) W)), so any slice can be regenerated and the injected sync words are known: every rank checks its
    hit count against that ground truth.  One step = one launch over the rank's 79 x (W / world) words."""
    NCH = 79
    W = int(args.gib * (1 << 30)) // 8 // NCH                     # words per channel
    ch_bits = W * 64 - 63                                         # offsets searched per channel
    plan = shard.plan(ch_bits, world)[rank]
    nw, nbits, fw = plan["n_words"], plan["search_bits"], plan["first_word"]
    assert nw > 0, "more ranks than words per channel"
    stream = torch.empty(NCH * nw, dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream(dev)
    hs = C.c_void_p(cur.cuda_stream)
    for c in range(NCH):                                          # row c = this rank's slice of channel c
        bt.check(lib.btbbx_synth_device(stream.data_ptr() + 8 * c * nw, c * W + fw, nw, SEED, STRIDE, -1, 4, hs))
    cap = NCH * (nbits // STRIDE + 64) + (1 << 16)
    hits_t = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    cnt_t = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        cnt_t.zero_()
        bt.check(lib.btbbx_scan_device(stream.data_ptr(), nw, nw, NCH, nbits, bt.LAP_ANY, 2, hits_t.data_ptr(), cap,
                                       cnt_t.data_ptr(), hs))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        cnt_t.zero_()
        ev[k][0].record(cur)
        bt.check(lib.btbbx_scan_device(stream.data_ptr(), nw, nw, NCH, nbits, bt.LAP_ANY, 2, hits_t.data_ptr(), cap,
                                       cnt_t.data_ptr(), hs))
        ev[k][1].record(cur)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kern_ms = sum(step_ms) / args.steps
    kern_median, kern_min = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]), step_ms[0]
    nhits = int(cnt_t.item())
    assert nhits <= cap, "hit buffer overflow"
    if args.dump_hits:
        mine_h = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:nhits].copy()
        mine_h["offset"] += np.uint64(plan["first_offset"])         # offset inside the channel (stream = channel number)
        np.save("%s.rank%d.npy" % (args.dump_hits, rank), mine_h)

    # ground truth: injected sync words with <= 2 bit errors whose window starts inside this rank's range of a channel
    from libbtbb_amd import synth
    true_hits = 0
    for c in range(NCH):
        lo = (c * W + fw) * 64
        k0, k1 = max(lo // STRIDE - 1, 0), (lo + nbits) // STRIDE + 1
        pos, _, nerr, mask = synth.injection_params(SEED, np.arange(k0, k1, dtype=np.uint64), STRIDE, 4)
        popc = np.unpackbits(mask.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1)
        true_hits += int(((popc <= 2) & (pos >= np.uint64(lo)) & (pos < np.uint64(lo + nbits))).sum())
    chance = 1.25e-8 * NCH * nbits
    hits_ok = 0 <= nhits - true_hits < 3 * chance + 100
    t_el = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=red_dev)
    sums = torch.tensor([float(NCH * nbits), float(nhits), float(true_hits), 1.0 if hits_ok else 0.0], dtype=torch.float64, device=red_dev)
    per_rank = [torch.zeros(4, dtype=torch.float64, device=red_dev) for _ in range(world)]
    mine = torch.tensor([float(rank), kern_ms, float(nhits), float(true_hits)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    if rank == 0:
        elapsed, kern_max = float(t_el[0]), float(t_el[1])
        total_bits = float(sums[0])
        alg_bytes = NCH * nbits / 8 + 16 * nhits                  # this rank's launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "Gbit/s raw bitstream scanned (LAP_ANY, err<=2)", "value": round(total_bits * args.steps / elapsed / 1e9, 2),
            "unit": "Gbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: promiscuous LAP_ANY over 79 channel streams, %.4g GiB packed in total, "
                                   "max_ac_errors=2, sync word every %d symbols, every channel time-sharded over %d rank(s)"
                                   % (args.gib, STRIDE, world),
                       "symbols_total": int(total_bits), "symbols_per_gpu": NCH * nbits, "hits_total": int(sums[1]),
                       "injected_hits_total": int(sums[2]), "hit_counts_match_ground_truth": bool(int(sums[3]) == world),
                       "per_rank": [{"rank": int(v[0]), "kernel_ms": round(float(v[1]), 4), "hits": int(v[2]),
                                     "injected_hits": int(v[3])} for v in per_rank],
                       "parallelism": "79 channels x time shards x%d (btbbx_shard_plan per channel), no collectives" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "scan_slide_kernel",
                         "kernel_ms": round(kern_ms, 4), "kernel_ms_max_over_ranks": round(kern_max, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
            "cpu_baseline": None,
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def ordered_headline(bt, lib, dev, cur, hs, stream, nwords, nbits, hits_t, cnt_t, cap, unordered_ms):
    """The headline scan with its hit list in (stream, offset) order -- what the host wrappers (btbbx_scan_host*, the streaming
    ingest) and btbb_find_ac hand out, and what the reference's loop produces by construction: btbbx_scan_ordered_device on the
    headline stream (the scan kernel counts every record in the bucket the ordering will put it in and parks its list in the
    scratch; scan of the counts, scatter, rank: sort.hip).  Checked in the run: strictly increasing offsets, and the same set
    of records as the unordered list of the timed headline loop."""
    unordered = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:int(cnt_t.item())].copy()
    order_bytes = lib.btbbx_scan_ordered_scratch_bytes(nbits, 1, bt.LAP_ANY, cap)     # (with the segment slots: 16 B per 4032 offsets)
    scratch = torch.empty(order_bytes, dtype=torch.uint8, device=dev)

    def step():
        bt.check(lib.btbbx_scan_ordered_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2, hits_t.data_ptr(), cap,
                                               cnt_t.data_ptr(), scratch.data_ptr(), order_bytes, hs))
    step()
    torch.cuda.synchronize()
    reps = 10

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        for _ in range(reps):
            fn()
        b.record(cur)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    # the unordered launch again, right beside the ordered one (the headline loop ran minutes ago, on a cooler chip): the ratio of
    # these two is what ordering costs
    def plain():
        cnt_t.zero_()
        bt.check(lib.btbbx_scan_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2, hits_t.data_ptr(), cap, cnt_t.data_ptr(), hs))
    plain()
    torch.cuda.synchronize()
    plain_ms = timed(plain)
    ms = timed(step)
    plain_ms = min(plain_ms, timed(plain))               # (before and behind the ordered loop)
    step()
    torch.cuda.synchronize()
    n = int(cnt_t.item())
    got = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:n].copy()
    increasing = bool(np.all(got["offset"][1:] > got["offset"][:-1]))
    full = ["offset", "lap", "ac_errors", "stream"]
    same = n == len(unordered) and bool(np.array_equal(got[full], np.sort(unordered, order="offset")[full]))
    alg = nbits / 8 + 16 * n + 32 * n                  # the stream once, every record written (slot), read and written again (compaction)
    del scratch
    return {
        "config": "BASELINE configs[1] with the list in offset order: btbbx_scan_ordered_device on the headline stream "
                  "(round 6: the scan leaves its hits, ranked, in slots of their 4032-offset segment; counts -> prefix -> one "
                  "coalesced copy), one stream, no host round trip",
        "scratch_bytes": int(order_bytes),
        "value": round(nbits / (ms * 1e-3) / 1e9, 2), "unit": "Gbit/s", "ms_per_step": round(ms, 4), "hits": n,
        "ordering_ms": round(ms - plain_ms, 4), "unordered_ms_beside": round(plain_ms, 4), "ordered_over_unordered": round(ms / plain_ms, 4),
        "unordered_kernel_ms_headline_loop": round(unordered_ms, 4),
        "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(alg),
                     "kernel": "scan_slide_kernel<ORD> + slot_*", "traffic": secondary_traffic("lap_any_4gib_ordered")[0],
                     "traffic_source": secondary_traffic("lap_any_4gib_ordered")[1],
                     "valu": valu_block(secondary_traffic("lap_any_4gib_ordered")[2], ms)},
        "parity": bool(increasing and same), "strictly_increasing": increasing, "equals_sorted_unordered_list": same,
    }


def four_error_headline(bt, lib, dev, cur, hs, stream, nwords, nbits, hits_t, cnt_t, cap):
    """The headline stream with tables for FOUR errors (btbb_init(4): 457 k error patterns) at max_ac_errors 4 -- scan_slide_kernel
    in its two-level form (a 2^20-bit set in LDS, its members looked up in a second set in L2).  Checked in the run: with the same
    tables and max_ac_errors 2 the list is the headline's own list (the tables only add patterns of more errors), and the list at 4
    contains it."""
    two = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:int(cnt_t.item())].copy()
    full = ["offset", "lap", "ac_errors", "stream"]
    cap = 2 * cap                                       # (a third more hits than with two errors)
    hits_t = torch.empty(2 * cap, dtype=torch.int64, device=dev)
    try:
        lib.btbbx_shutdown()
        bt.init(4)

        def step(me):
            cnt_t.zero_()
            bt.check(lib.btbbx_scan_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, me, hits_t.data_ptr(), cap,
                                           cnt_t.data_ptr(), hs))
        step(2)
        torch.cuda.synchronize()
        n2 = int(cnt_t.item())
        got2 = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:min(n2, cap)].copy()
        same = n2 == len(two) and bool(np.array_equal(np.sort(got2, order="offset")[full], np.sort(two, order="offset")[full]))
        step(4)
        torch.cuda.synchronize()
        reps = 5
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        for _ in range(reps):
            step(4)
        b.record(cur)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        n4 = int(cnt_t.item())
        fits = n4 <= cap
        got4 = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:min(n4, cap)]
        contains = fits and bool(np.isin(two["offset"], got4["offset"][got4["ac_errors"] <= 2]).all())
    finally:
        lib.btbbx_shutdown()
        bt.init(2)
    alg = nbits / 8 + 16 * n4
    return {
        "config": "BASELINE configs[1]'s stream with btbb_init(4) tables, max_ac_errors 4 (scan_slide_kernel, two-level form)",
        "value": round(nbits / (ms * 1e-3) / 1e9, 2), "unit": "Gbit/s", "ms_per_step": round(ms, 4), "hits": n4, "hits_at_2": n2,
        "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(alg),
                     "kernel": "scan_slide_kernel<Slide4>", "kernel_ms": round(ms, 4), "traffic": secondary_traffic("lap_any_4gib_init4")[0],
                     "traffic_source": secondary_traffic("lap_any_4gib_init4")[1],
                     "valu": valu_block(secondary_traffic("lap_any_4gib_init4")[2], ms)},
        "parity": bool(same and contains), "list_at_2_equals_headline": same, "list_at_4_contains_headline": contains,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib", type=float, default=None, help="packed stream size in GiB: per GPU for --layout single (default 4), "
                    "in total for --layout channels79 (default 64)")
    ap.add_argument("--cpu-symbols", type=int, default=0, help="size of the CPU-baseline sample (0 = 2^33 symbols when the host has >= 64 CPUs and the RAM for it, else 2^30)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 3 / config 5 block")
    ap.add_argument("--only-secondary", default=None, metavar="NAME", help="run one line of the secondary block only (with the input it "
                    "needs): lap_any_4gib_ordered, lap_any_4gib_init4, known_lap_79ch_chain_full_payloads, known_lap_79ch_chain, clk6_bruteforce, clk6_bruteforce_all_types")
    ap.add_argument("--layout", default="single", choices=["single", "channels79"],
                    help="single: one stream of --gib GiB per GPU (weak scaling, BASELINE configs[1], the default line); "
                         "channels79: BASELINE configs[3] as written -- 79 channel streams, --gib GiB IN TOTAL (default 64), every "
                         "channel time-sharded over the ranks (strong scaling)")
    ap.add_argument("--dump-hits", default=None, metavar="PREFIX", help="testing: every rank writes its hit list with GLOBAL "
                    "offsets to PREFIX.rank<r>.npy (tests/test_two_ranks.py merges them and compares with one scan)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "exercising the N>1 path where ranks share one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses cuda:0")
    args = ap.parse_args()

    if args.gib is None:
        args.gib = 4.0 if args.layout == "single" else 64.0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus must match WORLD_SIZE"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = pin_to_gpu_numa(local_rank) if world > 1 else None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    import libbtbb_amd as bt
    from libbtbb_amd import shard
    bt.init(2)
    lib = bt.lib()

    if args.layout == "channels79":
        return channels79(args, bt, lib, shard, dev, rank, world, red_dev)

    # the logical capture is `world` x --gib; this rank's shard of it (slice + 63-symbol halo)
    words_per_gpu = int(args.gib * (1 << 30)) // 8
    total_bits = world * words_per_gpu * 64 - 63
    plan = shard.plan(total_bits, world)[rank]
    nwords, nbits, first_word = plan["n_words"], plan["search_bits"], plan["first_word"]
    stream = torch.empty(nwords, dtype=torch.int64, device=dev)
    cap = nbits // STRIDE + (1 << 16)              # injections + chance matches + slack
    hits_t = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    cnt_t = torch.zeros(1, dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream(dev)
    hs = C.c_void_p(cur.cuda_stream)

    bt.check(lib.btbbx_synth_device(stream.data_ptr(), first_word, nwords, SEED, STRIDE, -1, 4, hs))
    torch.cuda.synchronize()

    def step():
        cnt_t.zero_()
        bt.check(lib.btbbx_scan_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2,
                                       hits_t.data_ptr(), cap, cnt_t.data_ptr(), hs))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        cnt_t.zero_()
        ev[k][0].record(cur)
        bt.check(lib.btbbx_scan_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2,
                                       hits_t.data_ptr(), cap, cnt_t.data_ptr(), hs))
        ev[k][1].record(cur)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kern_ms = sum(step_ms) / args.steps
    kern_median, kern_min = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]), step_ms[0]
    nhits = int(cnt_t.item())
    assert nhits <= cap, "hit buffer overflow"
    if args.dump_hits:
        mine_h = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:nhits].copy()
        mine_h["offset"] += np.uint64(plan["first_offset"])
        np.save("%s.rank%d.npy" % (args.dump_hits, rank), mine_h)
    t_el = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t_el[0]), float(t_el[1])

    result = None
    if rank == 0:
        value = total_bits * args.steps / elapsed / 1e9          # every offset of the logical capture, once per step
        alg_bytes = nbits / 8 + 16 * nhits
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # HBM traffic per launch: NOT measured in this run -- the value of the last rocprofv3 --pmc passes over
        # this same command (separate passes, FETCH_SIZE doubled as the guide prescribes), see the file named
        traffic, traffic_source, valu = None, None, None
        fp = csrc_fingerprint()
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if abs(args.gib - 4.0) < 1e-9 and tj.get("csrc_sha16") == fp:
                traffic = int(tj["bytes_per_launch"])
                v = tj.get("valu") or {}
                if v.get("insts_per_launch"):
                    # what the kernel is really bound by: fraction of the VALU-issue ceiling = wave-instructions x cycles per
                    # instruction / (SIMDs x clock) / kernel time (instruction count from the PMC run named in traffic_source)
                    busy_s = v["insts_per_launch"] * v["cycles_per_inst"] / (v["simds"] * v["clock_ghz"] * 1e9)
                    valu = {"insts_per_launch": int(v["insts_per_launch"]), "cycles_per_inst": v["cycles_per_inst"],
                            "issue_ms": round(busy_s * 1e3, 4), "frac": round(busy_s / (kern_ms * 1e-3), 4),
                            "pipe_busy_frac": v.get("pipe_busy_frac"), "pipe_cycles_per_inst": v.get("pipe_cycles_per_inst")}
                    # ... and how far the instruction stream itself is from the fewest vector instructions ANY exact filter of
                    # this shape needs per 64-bit stream word (NOTEBOOK.md 6.4: 7 barker planes + their adder tree for both halves
                    # 30, the sliding check stream 30, eight survivors x eight instructions since round 5: 64):
                    # kernel time x bound / measured = what this kernel would take at its own measured issue rate
                    per_word = v["insts_per_launch"] / (nwords / 64.0)
                    valu["bound"] = {"min_valu_per_word": VALU_MIN_PER_WORD, "measured_valu_per_word": round(per_word, 1),
                                     "frac": round(VALU_MIN_PER_WORD / per_word, 4),
                                     "bound_ms": round(kern_ms * VALU_MIN_PER_WORD / per_word, 4),
                                     "salu_per_valu": round((v.get("salu_insts_per_launch") or 0) / v["insts_per_launch"], 3)}
                traffic_source = "profiles/traffic.json (%s), same sources as this build (csrc_sha16 %s); not re-measured in this run" % (
                    tj.get("source", "rocprofv3 --pmc"), fp)
            else:
                traffic_source = "null: profiles/traffic.json was measured on sources %s, this build is %s" % (tj.get("csrc_sha16"), fp)
        except Exception:
            traffic = None
        result = {
            "metric": "Gbit/s raw bitstream scanned (LAP_ANY, err<=2)",
            "value": round(value, 2), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "promiscuous LAP_ANY scan, %.3g GiB packed single-channel synthetic "
                                   "bitstream per GPU, max_ac_errors=2, sync word every %d symbols"
                                   % (args.gib, STRIDE),
                       "symbols_per_gpu": nbits, "hits_per_gpu": nhits,
                       "parallelism": "time-sharded x%d (btbbx_shard_plan: slices + 63-symbol halo), no collectives" % world,
                       "rank0_host_pinning": numa},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "scan_slide_kernel", "kernel_ms": round(kern_ms, 4),
                         "kernel_ms_median": round(kern_median, 4), "kernel_ms_min": round(kern_min, 4), "kernel_ms_max": round(step_ms[-1], 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "valu": valu},
            "csrc_sha16": fp,
        }
        cpu = host_cpu()
        if world == 1 and not args.no_cpu:
            cpu_symbols = args.cpu_symbols
            if cpu_symbols <= 0:                     # >= 0.5 s per thread on a 256-thread host needs ~2^33 symbols (8 GiB of bytes)
                import psutil
                big = cpu["threads"] >= 64 and psutil.virtual_memory().available > (40 << 30)
                cpu_symbols = (1 << 33) if big else (1 << 30)
            ncpu_words = min(nwords, (cpu_symbols + 63) // 64)
            words_host = stream[:ncpu_words].cpu().numpy().view(np.uint64)
            raw = hits_t.cpu().numpy().view(bt.HIT_DTYPE)[:nhits]
            base, parity = cpu_baseline(words_host, 0, raw, cpu)
            result["cpu_baseline"] = base
            result["parity"] = bool(parity)
        else:
            result["cpu_baseline"] = None
        if world == 1 and not args.no_secondary:
            ordered = four = None
            if args.only_secondary in (None, "lap_any_4gib_init4"):
                four = four_error_headline(bt, lib, dev, cur, hs, stream, nwords, nbits, hits_t, cnt_t, cap)
            if args.only_secondary in (None, "lap_any_4gib_ordered"):
                if four is not None:                    # (the headline's own list back into hits_t)
                    cnt_t.zero_()
                    bt.check(lib.btbbx_scan_device(stream.data_ptr(), nwords, nwords, 1, nbits, bt.LAP_ANY, 2, hits_t.data_ptr(), cap,
                                                   cnt_t.data_ptr(), hs))
                    torch.cuda.synchronize()
                ordered = ordered_headline(bt, lib, dev, cur, hs, stream, nwords, nbits, hits_t, cnt_t, cap, kern_ms)
            del stream, hits_t
            torch.cuda.empty_cache()
            result["secondary"] = {} if args.only_secondary in ("lap_any_4gib_ordered", "lap_any_4gib_init4") else \
                secondary(bt, lib, dev, cur, hs, cpu, not args.no_cpu, args.only_secondary)
            if ordered is not None:
                result["secondary"]["lap_any_4gib_ordered"] = ordered
            if four is not None:
                result["secondary"]["lap_any_4gib_init4"] = four
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
