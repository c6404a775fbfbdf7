#!/bin/bash
# the whole GPU suite, one process, with per-test time limit
set -o pipefail
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/full_gpu.txt 2>&1
rc=$?
tail -5 gpurun_out/full_gpu.txt
exit $rc
