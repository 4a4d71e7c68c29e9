out=gpurun_out/r03_m; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_packets.py tests/test_baseline_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/tests.txt
tools/ab_trials.sh > $out/ab.txt 2>&1
cat $out/tests.txt $out/ab.txt
bash tools/kstats.sh r03_m decode_hits
