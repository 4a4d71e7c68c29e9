out=gpurun_out/r03_i; mkdir -p $out
export TMPDIR=/tmp
python tools/pmc_collect.py --out $out/pmc --kernel scan_ --groups FETCH_SIZE WRITE_SIZE \
  SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES \
  SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_INSTS_BRANCH \
  SQ_INST_CYCLES_VMEM,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_MISC \
  -- python bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $out/pmc_scan.json 2> $out/pmc.err
rm -rf $out/pmc
cat $out/pmc_scan.json | head -120; tail -5 $out/pmc.err
