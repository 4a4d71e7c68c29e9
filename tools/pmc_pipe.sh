#!/bin/bash
# Vector-pipe occupancy of every bench line's kernels: SQ_THREAD_CYCLES_VALU (lane-quad-cycles the pipe was busy) and SQ_CYCLES
# (summed over the 32 shader engines) in one --pmc pass per line -> <out>/pmc_pipe_<line>.json  (tools/collect_evidence.sh calls this;
# tools/make_traffic.py turns it into roofline.valu.pipe_busy_frac = THREAD_CYCLES_VALU / 64 x 4 / 1024 SIMDs / (SQ_CYCLES / 32))
out=${1:-gpurun_out/evidence}; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/pmc_collect.py --out $out/pmcp --kernel scan_ --groups SQ_THREAD_CYCLES_VALU,SQ_CYCLES,SQ_INSTS_VALU \
  -- python bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $out/pmc_pipe_headline.json 2> $out/pmcp.err
rm -rf $out/pmcp
for line in lap_any_4gib_ordered lap_any_4gib_init4 known_lap_79ch_chain_full_payloads known_lap_79ch_chain clk6_bruteforce clk6_bruteforce_all_types; do
  timeout 300 python tools/pmc_collect.py --out $out/pmcp --kernel "" --groups SQ_THREAD_CYCLES_VALU,SQ_CYCLES,SQ_INSTS_VALU \
    -- python bench.py --steps 2 --warmup 1 --no-cpu --only-secondary $line > $out/pmc_pipe_$line.json 2> $out/pmcp.err
  rm -rf $out/pmcp
done
rm -f $out/pmcp.err
