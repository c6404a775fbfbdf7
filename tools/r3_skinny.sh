#!/bin/bash
set -o pipefail
O=gpurun_out/r3l
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" --timeout 120 --timeout-method thread -p no:cacheprovider > $O/gemm_tests.log 2>&1 || { tail -40 $O/gemm_tests.log; exit 1; }
tail -2 $O/gemm_tests.log
timeout -k 10 200 python tools/gemm_skinny_bench.py > $O/gemm_skinny.txt 2>&1 || { tail $O/gemm_skinny.txt; exit 1; }
cat $O/gemm_skinny.txt
