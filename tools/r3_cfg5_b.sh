#!/bin/bash
set -o pipefail
O=gpurun_out/r3p
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bf16" --timeout 120 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
DVAE_GEMM_BF16=1 DVAE_GEMM_STORE=f32xbf16 timeout -k 10 100 python tools/gemm_bench.py > $O/gemm_shapes_f32xbf16.txt 2>&1 || { tail $O/gemm_shapes_f32xbf16.txt; exit 1; }
grep -E "cfg5|square|EB|EA" $O/gemm_shapes_f32xbf16.txt
timeout -k 10 200 python bench.py --config cfg5 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/cfg5_bf16.json 2> $O/cfg5_bf16.err || { tail -20 $O/cfg5_bf16.err; exit 1; }
timeout -k 10 200 python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg2_bf16.json 2> $O/cfg2_bf16.err || { tail -20 $O/cfg2_bf16.err; exit 1; }
python - <<PY
import json
for c in ("cfg5_bf16","cfg2_bf16"):
    d=json.load(open("$O/%s.json"%c))
    print(c, round(d["value"]), "DAGs/s", round(d["ms_per_step"],3), "ms/step")
PY
