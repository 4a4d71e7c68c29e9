#!/bin/bash
# The per-round evidence set of profiles/<dir> (run on the MI355X box from the repo root, on the COMMITTED build):
#   tools/collect_evidence.sh gpurun_out/r04_v1 ; then, in the build container: python tools/make_traffic.py profiles/r04_v1
#   csrc_sha16.txt    fingerprint of the sources measured (bench.csrc_fingerprint)
#   bench.json        the default `python bench.py` line
#   kernel_stats_<line>.csv  rocprofv3 --kernel-trace over ONE bench line at a time (headline: `bench.py --steps 10 --warmup 2
#                     --no-cpu --no-secondary`; the others `--only-secondary <line>`), a row per kernel, warm-up launches dropped
#   pmc_scan.json     separate --pmc passes (never combined with traces) over the headline loop: FETCH_SIZE, WRITE_SIZE, SQ_*
#   pmc_sec_<line>.json  the same for ONE line of the secondary block at a time (bench.py --only-secondary <line>): a kernel's
#                     mean per launch then belongs to one workload (the two config-3 captures share their kernels)
#   pmc_pipe_<line>.json  SQ_THREAD_CYCLES_VALU / SQ_CYCLES per kernel: how busy the vector pipe was (tools/pmc_pipe.sh)
#   init_sweep.json, multistream.txt, decode_by_type.txt
out=${1:-gpurun_out/evidence}
mkdir -p $out
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_fingerprint())" > $out/csrc_sha16.txt
timeout 300 python bench.py 2>/dev/null | tail -1 > $out/bench.json
# per-kernel durations: one file per bench line, warm-up launches dropped (tools/kstats_by_line.py)
timeout 900 python tools/kstats_by_line.py $out > $out/kstats.log 2>&1
timeout 300 python tools/pmc_collect.py --out $out/pmc --kernel scan_ --groups FETCH_SIZE WRITE_SIZE \
  SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES \
  SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY \
  -- python bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $out/pmc_scan.json 2> $out/pmc.err
rm -rf $out/pmc
for line in lap_any_4gib_ordered lap_any_4gib_init4 known_lap_79ch_chain_full_payloads known_lap_79ch_chain clk6_bruteforce clk6_bruteforce_all_types; do
  timeout 300 python tools/pmc_collect.py --out $out/pmc2 --kernel "" --groups FETCH_SIZE WRITE_SIZE \
    SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY \
    -- python bench.py --steps 2 --warmup 1 --no-cpu --only-secondary $line > $out/pmc_sec_$line.json 2> $out/pmc2.err
  rm -rf $out/pmc2
done
tools/pmc_pipe.sh $out
timeout 120 python tools/init_sweep.py 2>/dev/null | tail -1 > $out/init_sweep.json
timeout 120 python tools/multistream_time.py 2>/dev/null > $out/multistream.txt
timeout 200 python tools/decode_time.py > $out/decode_by_type.txt 2>/dev/null
