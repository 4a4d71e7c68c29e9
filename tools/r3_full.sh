#!/bin/bash
# full GPU check of the current build: pytest -m gpu, default bench line, soak
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $out/pytest_gpu.txt
python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
timeout 400 python tools/soak.py --seconds ${2:-120} --seed ${3:-5} 2>&1 | tail -3 > $out/soak.txt
cat $out/pytest_gpu.txt $out/soak.txt; cat $out/bench.json; tail -3 $out/bench.err
