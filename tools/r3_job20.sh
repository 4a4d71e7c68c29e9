out=gpurun_out/r03_p; mkdir -p $out
export TMPDIR=/tmp
python tools/pmc_collect.py --out $out/pmc --kernel scan_known --groups \
  SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY \
  SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_BRANCH,SQ_WAVES SQ_INST_CYCLES_VMEM,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS,SQ_BUSY_CYCLES \
  -- python bench.py --steps 2 --warmup 1 --no-cpu > $out/pmc_known.json 2> $out/pmc.err
rm -rf $out/pmc
cat $out/pmc_known.json | head -5; tail -2 $out/pmc.err
