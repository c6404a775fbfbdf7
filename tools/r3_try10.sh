#!/bin/bash
set -o pipefail
O=gpurun_out/r3n
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -2 $O/parity.log
timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err || { tail -20 $O/cfg1.err; exit 1; }
timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err || { tail -20 $O/cfg2.err; exit 1; }
python - <<PY
import json
for c in ("cfg1","cfg2"):
    d=json.load(open("$O/%s.json"%c))
    print(c, round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "unblocked", round(d["config"]["host_enqueue_ms_per_step_unblocked"],3), "launches", d["config"]["launches_per_step"])
PY
timeout -k 10 120 python tools/host_cost.py cfg1 20 > $O/host_cost.txt 2>&1 && grep "host enq" $O/host_cost.txt
bash tools/timeline.sh $O/timeline_cfg1.txt --config cfg1 --steps 10 --warmup 5
grep -E "vertex_ce|pair_fwd" $O/timeline_cfg1.txt
