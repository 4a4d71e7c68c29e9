#!/usr/bin/env python3
"""profiles/traffic.json and profiles/traffic_secondary.json from the PMC passes of tools/collect_evidence.sh:
    python tools/make_traffic.py profiles/r03_v1
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (the gfx950 correction of MI355X_MICROARCH.md's HBM
section, calibrated on this kernel's load shape in profiles/r01_calib).  Both files carry csrc_sha16 = the fingerprint
of the sources that were measured (bench.csrc_fingerprint, written by collect_evidence.sh into <dir>/csrc_sha16.txt);
bench.py prints traffic: null when the build it runs differs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
rel = os.path.relpath(d, ROOT)
fp = open(os.path.join(d, "csrc_sha16.txt")).read().strip()
bench = json.load(open(os.path.join(d, "bench.json")))


def hbm(entry):
    return (2 * entry["FETCH_SIZE"]["mean_per_launch"] + entry["WRITE_SIZE"]["mean_per_launch"]) * 1024


def pipe(path, kernel_substr):
    """measured occupancy of the vector pipe by the kernel whose name contains kernel_substr (tools/pmc_pipe.sh): SQ_THREAD_CYCLES_VALU
    counts lane-quad-cycles, SQ_CYCLES is summed over the 32 shader engines -> fraction of the 1024 SIMDs' cycles, cycles per instruction"""
    try:
        e = json.load(open(path))
    except Exception:
        return {}
    ks = [k for k in e if kernel_substr in k and "SQ_THREAD_CYCLES_VALU" in e[k] and "SQ_CYCLES" in e[k]]
    if not ks:
        return {}
    k = max(ks, key=lambda k: e[k]["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"])
    busy = e[k]["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / 64.0 * 4.0
    cyc = e[k]["SQ_CYCLES"]["mean_per_launch"] / 32.0
    r = {"pipe_busy_frac": round(busy / 1024.0 / cyc, 3)}
    if e[k].get("SQ_INSTS_VALU"):
        r["pipe_cycles_per_inst"] = round(busy / e[k]["SQ_INSTS_VALU"]["mean_per_launch"], 2)
    return r


pmc = json.load(open(os.path.join(d, "pmc_scan.json")))
name = [k for k in pmc if "scan_slide_kernel" in k][0]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
b = hbm(pmc[name])
json.dump({
    "bytes_per_launch": int(b), "csrc_sha16": fp,
    "source": "%s/pmc_scan.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 4 "
              "--warmup 1 --no-cpu` (tools/collect_evidence.sh)" % rel,
    "derivation": "(2 x FETCH_SIZE + WRITE_SIZE) KB x 1024; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for "
                  "gfx950 and as calibrated on this kernel's own load shape (profiles/r01_calib: 0.5105 of the true bytes) and on the packet kernels' shapes (profiles/r05_calib: every fabric request is a whole 128-byte line, tallied at 64)",
    "FETCH_SIZE_KB": pmc[name]["FETCH_SIZE"]["mean_per_launch"], "WRITE_SIZE_KB": pmc[name]["WRITE_SIZE"]["mean_per_launch"],
    "algorithmic_bytes_per_launch": alg, "ratio_to_algorithmic": round(b / alg, 3), "kernel": name + ", 4 GiB LAP_ANY bench workload",
    # the VALU-issue ceiling bench.py quotes next to the HBM roofline (roofline.valu): wave-instructions per launch from the same
    # PMC run; cycles per instruction = the kernel's static mix (28 % of its VALU instructions are half-rate ones: v_alignbit,
    # v_ffbl, v_mbcnt ...) priced with tools/valu_rate.bin (profiles/r03_ab/valu_rate.txt: 2.65 / 4.35 cycles at 2.4 GHz)
    "valu": {"insts_per_launch": pmc[name].get("SQ_INSTS_VALU", {}).get("mean_per_launch"), "cycles_per_inst": 3.1,
             "simds": 1024, "clock_ghz": 2.4,
             "lds_insts_per_launch": pmc[name].get("SQ_INSTS_LDS", {}).get("mean_per_launch"),
             "salu_insts_per_launch": pmc[name].get("SQ_INSTS_SALU", {}).get("mean_per_launch"),
             **pipe(os.path.join(d, "pmc_pipe_headline.json"), "scan_slide_kernel")},
}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("traffic.json: %.3f GB per launch = %.3f x algorithmic" % (b / 1e9, b / alg))

out = {"csrc_sha16": fp, "source": "%s/pmc_sec_<line>.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
       "`python bench.py --steps 2 --warmup 1 --no-cpu --only-secondary <line>` (one line of the block at a time: a kernel's mean "
       "per launch then belongs to one workload); (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch, as in traffic.json" % rel}
# (round 6: the ordered lines run the scan kernels' slot form -- "..., true>" -- and the compaction; the plain forms also appear in those
# runs, as the lines' own unordered timing and as the gated fallback's launches that return at once: not part of the step)
CHAIN = ("scan_known_lap_kernel<2, 1, false, true>", "slot_", "order_single", "decode_hits_kernel")
DOMINANT = {"lap_any_4gib_ordered": "scan_slide_kernel<SlideStd, 2, false, true>", "lap_any_4gib_init4": "Slide4",
            "known_lap_79ch_chain_full_payloads": "scan_known_lap_kernel<2, 1, false, true>", "known_lap_79ch_chain": "scan_known_lap_kernel<2, 1, false, true>", "clk6_bruteforce": "trials_linear_kernel", "clk6_bruteforce_all_types": "trials_linear_kernel"}
for line, keys in (("lap_any_4gib_ordered", ("scan_slide_kernel<SlideStd, 2, false, true>", "slot_", "order_single")), ("lap_any_4gib_init4", ("Slide4",)), ("known_lap_79ch_chain_full_payloads", CHAIN), ("known_lap_79ch_chain", CHAIN),
                   ("clk6_bruteforce", ("trials_linear_kernel", "trials_wave_kernel")),
                   ("clk6_bruteforce_all_types", ("trials_linear_kernel", "trials_wave_kernel"))):
    path = os.path.join(d, "pmc_sec_%s.json" % line)
    if not os.path.exists(path) or line not in bench.get("secondary", {}):
        continue
    sec = json.load(open(path))
    used = sorted(k for k in sec if any(x in k for x in keys) and "FETCH_SIZE" in sec[k] and "WRITE_SIZE" in sec[k])
    if not used:
        continue
    tot = sum(hbm(sec[k]) for k in used)
    alg = bench["secondary"][line]["roofline"]["algorithmic_bytes_per_step"]
    out[line] = {"bytes_per_step": int(tot), "kernels": " + ".join(used), "algorithmic_bytes_per_step": alg,
                 "ratio_to_algorithmic": round(tot / alg, 3),
                 "per_kernel_bytes": {k: int(hbm(sec[k])) for k in used}}
    # the line's dominant kernel for roofline.valu (bench.valu_block): instruction counts of the same PMC passes; cycles per
    # instruction = 3.1, the scan kernels' static mix priced with tools/valu_rate.hip (2.4 / 4.2 cycles for full- and half-rate forms)
    dom = [k for k in used if DOMINANT[line] in k]
    if dom and sec[dom[0]].get("SQ_INSTS_VALU"):
        e = sec[dom[0]]
        wc = e.get("SQ_WAVE_CYCLES", {}).get("mean_per_launch")
        out[line]["valu"] = {"kernel": dom[0], "insts_per_launch": e["SQ_INSTS_VALU"]["mean_per_launch"], "cycles_per_inst": 3.1,
                             "simds": 1024, "clock_ghz": 2.4,
                             "salu_insts_per_launch": e.get("SQ_INSTS_SALU", {}).get("mean_per_launch"),
                             "lds_insts_per_launch": e.get("SQ_INSTS_LDS", {}).get("mean_per_launch"),
                             "wait_any_frac": round(e["SQ_WAIT_ANY"]["mean_per_launch"] / wc, 3) if wc and e.get("SQ_WAIT_ANY") else None,
                             **pipe(os.path.join(d, "pmc_pipe_%s.json" % line), DOMINANT[line])}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_secondary.json"), "w"), indent=1)
print("traffic_secondary.json:", {k: v["ratio_to_algorithmic"] for k, v in out.items() if isinstance(v, dict)})
