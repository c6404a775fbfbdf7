#!/bin/bash
set -o pipefail
O=gpurun_out/r3q
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -x -q -m gpu -k "cfg5 or pair" --timeout 300 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for pb in 1024 1536 1792 2048; do
  DVAE_PAIR_BLOCKS=$pb timeout -k 10 200 python tools/phase_times.py cfg5 3 bf16 > $O/phase_$pb.txt 2>&1 || { tail $O/phase_$pb.txt; exit 1; }
  echo "== DVAE_PAIR_BLOCKS=$pb"; grep -v amdgpu $O/phase_$pb.txt | tr '\n' ';'; echo
done
