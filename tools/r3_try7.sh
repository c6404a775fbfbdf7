#!/bin/bash
set -o pipefail
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or random_graphs or edge_cases or n64" --timeout 120 --timeout-method thread -p no:cacheprovider > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -2 $O/parity.log
timeout -k 10 200 python tools/tiny_prof.py > $O/tiny_prof.txt 2>&1 || { tail $O/tiny_prof.txt; exit 1; }
tail -14 $O/tiny_prof.txt
timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err || { tail -20 $O/cfg1.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/cfg1.json"))
print(round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "launches", d["config"]["launches_per_step"])
PY
timeout -k 10 120 python tools/host_cost.py cfg1 20 > $O/host_cost.txt 2>&1 && cat $O/host_cost.txt
