out=gpurun_out/r03_h; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_sanitizers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/tests.txt
tools/ab_trials.sh > $out/ab_trials.txt 2>&1
cat $out/tests.txt $out/ab_trials.txt
