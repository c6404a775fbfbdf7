#!/bin/bash
# round 3: bf16-stored d EA / d EB (DVAE_DTABLES_BF16) -- tests, then the cfg 5 bf16 step with and without
set -o pipefail
timeout -k 10 500 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests/test_gpu_ops.py -k "stored_kmajor or wgrad_splitk or k_chunks or segment_colsum or pair" > gpurun_out/h_ops.txt 2>&1 || { tail -30 gpurun_out/h_ops.txt; exit 1; }
tail -2 gpurun_out/h_ops.txt
timeout -k 10 600 python -u -m pytest -x -q --timeout 240 --timeout-method thread tests/test_gpu_parity.py tests/test_gpu_chain_modes.py -k "cfg5 or bf16" > gpurun_out/h_par.txt 2>&1 || { tail -30 gpurun_out/h_par.txt; exit 1; }
tail -2 gpurun_out/h_par.txt
for ab in 0 1; do
  DVAE_DTABLES_BF16=$ab timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/h_bench$ab.txt 2>&1 || { tail gpurun_out/h_bench$ab.txt; exit 1; }
  echo "== DVAE_DTABLES_BF16=$ab"; python - <<PY
import json
l=[x for x in open("gpurun_out/h_bench$ab.txt") if x.startswith("{")][-1]
j=json.loads(l); print(j["value"], j["ms_per_step"])
PY
done
