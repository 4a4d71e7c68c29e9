#!/bin/bash
# The per-round evidence set of profiles/<dir> (run on the MI355X box from the repo root):
#   tools/collect_evidence.sh gpurun_out/r02_v7
# bench.json (default bench line), kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_scan.json (separate
# --pmc passes, never combined with traces), init_sweep.json, multistream.txt
out=${1:-gpurun_out/evidence}
mkdir -p $out
export TMPDIR=/tmp
python bench.py 2>/dev/null | tail -1 > $out/bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1 )
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
python tools/pmc_collect.py --out $out/pmc --kernel scan_ --groups FETCH_SIZE WRITE_SIZE \
  SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES \
  SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY \
  -- python bench.py --steps 4 --warmup 1 --no-cpu > $out/pmc_scan.json 2> $out/pmc.err
rm -rf $out/pmc
python tools/init_sweep.py 2>/dev/null | tail -1 > $out/init_sweep.json
python tools/multistream_time.py 2>/dev/null > $out/multistream.txt
