#!/bin/bash
set -o pipefail
O=gpurun_out/r3g
mkdir -p $O
for side in 1 0; do for g in on off; do
  DVAE_SIDE_STREAMS=$side timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline --graph $g > $O/cfg1_side${side}_graph_$g.json 2> $O/cfg1_side${side}_graph_$g.err || { tail -20 $O/cfg1_side${side}_graph_$g.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg1_side${side}_graph_$g.json"))
c=d["config"]
print("side=$side graph=$g", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(c["host_enqueue_ms_per_step"],3), "unblocked", round(c["host_enqueue_ms_per_step_unblocked"],3), c["graph"])
PY
done; done
