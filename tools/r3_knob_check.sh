#!/bin/bash
# the A/B knobs of the bf16 storage work must each leave a working step (cfg 5 bf16, 2 steps)
set -o pipefail
for kv in DVAE_BF16_WEIGHTS=0 DVAE_TABLES_BF16=0 DVAE_DTABLES_BF16=0 DVAE_ACT_BF16=0 DVAE_GATES_BF16=0 DVAE_GRADS_BF16=0 DVAE_SIDE_STREAMS=0; do
  env $kv timeout -k 10 300 python bench.py --config cfg5 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/knob.txt 2>&1 || { echo "$kv FAILED"; tail -5 gpurun_out/knob.txt; exit 1; }
  python - "$kv" <<'PY'
import json, sys
l=[x for x in open("gpurun_out/knob.txt") if x.startswith("{")][-1]
j=json.loads(l); print(sys.argv[1], round(j["value"]), round(j["ms_per_step"], 2))
PY
done
