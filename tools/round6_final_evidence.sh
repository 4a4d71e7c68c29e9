set -x
out=gpurun_out/r06_v11
mkdir -p $out
tools/full_check.sh r06_v11 120 51 > $out/full_check.log 2>&1
cp $out/bench.json $out/bench_with_cpu_baseline.json
tools/collect_evidence.sh $out > $out/collect.log 2>&1
timeout 300 python -m pytest tests/test_two_ranks.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 > $out/eight_ranks.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-gpu --no-cpu 2>/dev/null | tail -1 > $out/two_ranks_bench.txt
BTBB_TEST_SEED=63 timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 > $out/pytest_gpu_seed_63.txt
ls -la $out | head -50
cat $out/full_check.log | tail -12
