#!/bin/bash
set -o pipefail
O=gpurun_out/r3v
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_chain_modes.py tests/test_gpu_parity.py -x -q -m gpu -k "shadows or cfg5 or bf16_mode" --timeout 600 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for ab in 0 1; do
  DVAE_ACT_BF16=$ab timeout -k 10 200 python tools/phase_times.py cfg5 3 bf16 > $O/phase_$ab.txt 2>&1 || { tail $O/phase_$ab.txt; exit 1; }
  echo "== DVAE_ACT_BF16=$ab"; grep -v amdgpu $O/phase_$ab.txt | tr '\n' ';'; echo
  DVAE_ACT_BF16=$ab timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/cfg5_$ab.json 2> $O/cfg5_$ab.err || { tail -20 $O/cfg5_$ab.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/cfg5_$ab.json"))
print("DVAE_ACT_BF16=$ab", round(d["value"]), "DAGs/s", round(d["ms_per_step"],2), "ms/step loss", d["config"].get("loss_last_step"))
PY
done
