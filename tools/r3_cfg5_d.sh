#!/bin/bash
set -o pipefail
O=gpurun_out/r3t
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -x -q -m gpu -k "cfg5 or pair or bf16" --timeout 300 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python - <<PY
import json
d=json.load(open("gpurun_out/parity/bf16_errors_dvae_cfg5_bi.json"))
print("cfg5-shape bf16: loss", d["loss_rel"], "mu", d["mu_rel"], "worst grad", max(d["grad_rel"].items(), key=lambda kv: kv[1]))
PY
for tb in 0 1; do
  DVAE_TABLES_BF16=$tb timeout -k 10 200 python tools/phase_times.py cfg5 3 bf16 > $O/phase_$tb.txt 2>&1 || { tail $O/phase_$tb.txt; exit 1; }
  echo "== DVAE_TABLES_BF16=$tb"; grep -v amdgpu $O/phase_$tb.txt | tr '\n' ';'; echo
  DVAE_TABLES_BF16=$tb timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/cfg5_$tb.json 2> $O/cfg5_$tb.err || { tail -20 $O/cfg5_$tb.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg5_$tb.json"))
print("DVAE_TABLES_BF16=$tb", round(d["value"]), "DAGs/s", round(d["ms_per_step"],2), "ms/step loss", d["config"].get("loss_last_step"))
PY
done
