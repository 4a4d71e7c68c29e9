out=gpurun_out/r03_g; mkdir -p $out
timeout 900 python -m pytest tests/test_two_ranks.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/two_ranks.txt
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1 )
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
cat $out/two_ranks.txt; head -40 $out/kernel_stats.csv | cut -c1-150
python bench.py --layout channels79 --gib 8 --steps 5 --warmup 1 2>/dev/null | tail -1 > $out/ch79_8gib.json; cut -c1-1200 $out/ch79_8gib.json
