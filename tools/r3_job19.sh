out=gpurun_out/r03_o; mkdir -p $out
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/kl4.so; do
LIBBTBB_AMD_SO=$PWD/$so timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_baseline_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 >> $out/tests.txt
done
tools/ab_trials.sh > $out/ab.txt 2>&1
cat $out/tests.txt; cut -c1-110 $out/ab.txt
