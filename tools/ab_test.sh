#!/bin/bash
# correctness of a kernel build before it is timed: the GPU scan tests against one .so
so=$1
LIBBTBB_AMD_SO=$PWD/$so timeout 600 python -m pytest tests/test_gpu_scan.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
