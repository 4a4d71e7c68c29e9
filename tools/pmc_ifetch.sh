#!/bin/bash
# instruction-fetch side counters of the headline kernel for the normal build and the listed variants
out=gpurun_out/$1; mkdir -p $out; shift
for so in libbtbb_amd/libbtbb_amd.so "$@"; do
  name=$(basename $so .so)
  LIBBTBB_AMD_SO=$PWD/$so timeout 600 python tools/pmc_collect.py --out $out/pmc_$name --kernel scan_slide --groups \
    SQ_IFETCH,SQ_IFETCH_LEVEL,SQ_INSTS_BRANCH,SQ_CYCLES \
    SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,SQC_ICACHE_BUSY_CYCLES \
    SQ_ACTIVE_INST_SCA,SQ_INST_CYCLES_SALU,SQ_THREAD_CYCLES_VALU,SQ_WAIT_INST_LDS \
    SQC_ICACHE_INPUT_VALID_READYB,SQC_TC_INST_REQ,SQ_BUSY_CU_CYCLES,SQ_WAVE_CYCLES \
    SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU \
    -- python bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $out/pmc_if_$name.json 2> $out/pmc_if_$name.err
  rm -rf $out/pmc_$name
done
python - $out <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/pmc_if_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k, v in d.items():
        print(os.path.basename(f), k[:50])
        print("   ", {c: round(x["mean_per_launch"] / 1e6, 2) for c, x in sorted(v.items())})
PY
