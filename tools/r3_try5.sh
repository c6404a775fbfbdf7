#!/bin/bash
set -o pipefail
O=gpurun_out/r3f
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_harness.py -x -q -m gpu -k "graphed" --timeout 200 --timeout-method thread -p no:cacheprovider > $O/graphed.log 2>&1 || { tail -60 $O/graphed.log; exit 1; }
tail -3 $O/graphed.log
for g in on off; do
  timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline --graph $g > $O/cfg1_graph_$g.json 2> $O/cfg1_graph_$g.err || { tail -20 $O/cfg1_graph_$g.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg1_graph_$g.json"))
c=d["config"]
print("graph=$g", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(c["host_enqueue_ms_per_step"],3), "unblocked", round(c["host_enqueue_ms_per_step_unblocked"],3), "launches", c["launches_per_step"], c["graph"], "loss", c["loss_last_step"])
PY
done
for g in on off; do
  timeout -k 10 120 python bench.py --config cfg2 --steps 50 --warmup 8 --no-cpu-baseline --graph $g > $O/cfg2_graph_$g.json 2> $O/cfg2_graph_$g.err || { tail -20 $O/cfg2_graph_$g.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg2_graph_$g.json"))
c=d["config"]
print("cfg2 graph=$g", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(c["host_enqueue_ms_per_step"],3), c["graph"], "loss", c["loss_last_step"])
PY
done
