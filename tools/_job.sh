timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_baseline_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
for so in libbtbb_amd/libbtbb_amd.so libbtbb_amd/variants/prev.so; do echo $so; LIBBTBB_AMD_SO=$PWD/$so python tools/known_lap_time.py 2>&1 | grep "4 GiB"; done
tools/ab_trials.sh 2>&1 | cut -c1-100
