#!/bin/bash
# full GPU check of the current build: pytest -m gpu, default bench line, soak
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > $out/pytest_gpu.txt
python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
[ "${2:-0}" != "0" ] && timeout 400 python tools/soak.py --seconds $2 --seed ${3:-5} 2>&1 | tail -3 > $out/soak.txt
cat $out/pytest_gpu.txt $out/soak.txt 2>/dev/null; python - <<PY
import json
d=json.load(open("$out/bench.json"))
print("headline", d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], "parity", d.get("parity"))
for k,v in (d.get("secondary") or {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v["roofline"].get("frac"), v["roofline"].get("kernel_ms"), "parity", v.get("parity"))
PY
tail -3 $out/bench.err
