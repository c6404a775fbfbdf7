#!/bin/bash
# round 3, first GPU try of the persistent small-batch chains: parity, then cfg 1 with and without them
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/r3a/parity.log 2>&1 || { tail -30 gpurun_out/r3a/parity.log; exit 1; }
tail -3 gpurun_out/r3a/parity.log
timeout -k 10 400 python -u -m pytest tests/test_gpu_chain_modes.py -x -q -m gpu -k "tiny" --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r3a/modes.log 2>&1 || { tail -30 gpurun_out/r3a/modes.log; exit 1; }
tail -3 gpurun_out/r3a/modes.log
for p in 1 0; do
  DVAE_CHAIN_TINY_PERSIST=$p timeout -k 10 120 python bench.py --config cfg1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3a/cfg1_persist$p.json 2> gpurun_out/r3a/cfg1_persist$p.err || { tail -20 gpurun_out/r3a/cfg1_persist$p.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3a/cfg1_persist$p.json"))
print("persist=$p", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "launches", d["config"]["launches_per_step"], "loss", d["config"]["loss_last_step"])
PY
done
DVAE_CHAIN_TINY_PERSIST=1 timeout -k 10 120 python tools/host_cost.py cfg1 20 > gpurun_out/r3a/host_cost_persist1.txt 2>&1 && cat gpurun_out/r3a/host_cost_persist1.txt
DVAE_CHAIN_TINY_PERSIST=0 timeout -k 10 120 python tools/host_cost.py cfg1 20 > gpurun_out/r3a/host_cost_persist0.txt 2>&1 && cat gpurun_out/r3a/host_cost_persist0.txt
