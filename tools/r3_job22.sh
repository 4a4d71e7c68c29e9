out=gpurun_out/r03_q; mkdir -p $out
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/nt.so; do
  echo "$so $(LIBBTBB_AMD_SO=$PWD/$so SWEEP_N=2,3,4,5 python tools/init_sweep.py 2>/dev/null | tail -1)" >> $out/sweep.txt
done
cat $out/sweep.txt
