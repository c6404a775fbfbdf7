#!/bin/bash
set -o pipefail
O=gpurun_out/r3o
mkdir -p $O
for rows in 64 128; do
  DVAE_CHAIN_BF16_ROWS=$rows timeout -k 10 300 python -u -m pytest tests/test_gpu_chain_bf16.py -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > $O/chain_bf16_tests_$rows.log 2>&1 || { tail -40 $O/chain_bf16_tests_$rows.log; exit 1; }
  tail -1 $O/chain_bf16_tests_$rows.log
  DVAE_CHAIN_BF16_ROWS=$rows timeout -k 10 200 python tools/chain_bf16_bench.py > $O/chain_bf16_levels_$rows.txt 2>&1 || { tail $O/chain_bf16_levels_$rows.txt; exit 1; }
  echo "== rows $rows"; grep -v amdgpu.ids $O/chain_bf16_levels_$rows.txt
done
for mode in "0 64" "1 64" "1 128"; do
  set -- $mode
  DVAE_CHAIN_BF16=$1 DVAE_CHAIN_BF16_ROWS=$2 timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/cfg5_bf16_$1_$2.json 2> $O/cfg5_bf16_$1_$2.err || { tail -20 $O/cfg5_bf16_$1_$2.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg5_bf16_$1_$2.json"))
print("DVAE_CHAIN_BF16=$1 ROWS=$2", round(d["value"]), "DAGs/s", round(d["ms_per_step"],2), "ms/step")
PY
done
