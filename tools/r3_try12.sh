#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest -x -q --timeout 240 --timeout-method thread tests/test_gpu_parity.py tests/test_gpu_harness.py > gpurun_out/t12_par.txt 2>&1 || { tail -30 gpurun_out/t12_par.txt; exit 1; }
tail -2 gpurun_out/t12_par.txt
for cfg in cfg1 cfg2; do
  st=50; [ $cfg = cfg1 ] && st=300
  timeout -k 10 200 python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline > gpurun_out/t12_$cfg.txt 2>&1 || { tail gpurun_out/t12_$cfg.txt; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/t12_$cfg.txt") if x.startswith("{")][-1]
j=json.loads(l); print("$cfg", j["value"], j["ms_per_step"], j["config"].get("launches_per_step"))
PY
done
