#!/bin/bash
set -o pipefail
O=gpurun_out/r3s
mkdir -p $O
for g in 1 2 3 4 1 2; do
  DVAE_WBL_GROUP=$g timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg2_g$g.json 2> $O/cfg2_g$g.err || { tail -20 $O/cfg2_g$g.err; exit 1; }
  DVAE_WBL_GROUP=$g timeout -k 10 200 python bench.py --config cfg4 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg4_g$g.json 2> $O/cfg4_g$g.err || { tail -20 $O/cfg4_g$g.err; exit 1; }
  python - <<PY
import json
a=json.load(open("$O/cfg2_g$g.json")); b=json.load(open("$O/cfg4_g$g.json"))
print("DVAE_WBL_GROUP=$g cfg2", round(a["value"]), round(a["ms_per_step"],4), "launches", a["config"]["launches_per_step"], " cfg4", round(b["value"]), round(b["ms_per_step"],4))
PY
done
DVAE_WBL_GROUP=2 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch or hs501" --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -2
