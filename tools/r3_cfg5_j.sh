#!/bin/bash
# round 3: half grads (DVAE_GRADS_BF16) -- op tests, forced-mode parity, then the cfg 5 bf16 step with and without
set -o pipefail
timeout -k 10 500 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests/test_gpu_ops.py -k "stored_kmajor or wgrad_splitk or k_chunks or colsum or pointwise or gated" > gpurun_out/j_ops.txt 2>&1 || { tail -30 gpurun_out/j_ops.txt; exit 1; }
tail -2 gpurun_out/j_ops.txt
timeout -k 10 900 python -u -m pytest -x -q --timeout 600 --timeout-method thread tests/test_gpu_chain_modes.py -k "activation_shadows" > gpurun_out/j_par.txt 2>&1 || { tail -40 gpurun_out/j_par.txt; exit 1; }
tail -2 gpurun_out/j_par.txt
cp gpurun_out/parity/bf16_errors_dvae_cfg5_bi.json gpurun_out/j_bf16_errors_gh.json 2>/dev/null
for ab in 0 1; do
  DVAE_GRADS_BF16=$ab timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/j_bench$ab.txt 2>&1 || { tail gpurun_out/j_bench$ab.txt; exit 1; }
  echo "== DVAE_GRADS_BF16=$ab"; python - <<PY
import json
l=[x for x in open("gpurun_out/j_bench$ab.txt") if x.startswith("{")][-1]
j=json.loads(l); print(j["value"], j["ms_per_step"])
PY
done
