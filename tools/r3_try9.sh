#!/bin/bash
set -o pipefail
O=gpurun_out/r3m
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_chain_modes.py -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for mm in 0 160; do
DVAE_GEMM_SKINNY_MMAC=$mm timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline > $O/cfg1_$mm.json 2> $O/cfg1_$mm.err || { tail -20 $O/cfg1_$mm.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/cfg1_$mm.json"))
print("MMAC=$mm", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "unblocked", round(d["config"]["host_enqueue_ms_per_step_unblocked"],3), "launches", d["config"]["launches_per_step"])
PY
DVAE_GEMM_SKINNY_MMAC=$mm timeout -k 10 120 python tools/host_cost.py cfg1 20 > $O/host_cost_$mm.txt 2>&1 && grep "host enq" $O/host_cost_$mm.txt
done
for mm in 0 160; do
DVAE_GEMM_SKINNY_MMAC=$mm timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/cfg2_$mm.json 2> $O/cfg2_$mm.err || { tail -20 $O/cfg2_$mm.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/cfg2_$mm.json"))
print("cfg2 MMAC=$mm", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step")
PY
done
