#!/bin/bash
set -o pipefail
O=gpurun_out/r3r
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py tests/test_gpu_dist_two_rank.py tests/test_gpu_dist_rccl.py -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 200 python tools/host_profile.py cfg1 300 > $O/host_profile_cfg1.txt 2>&1 || { tail $O/host_profile_cfg1.txt; exit 1; }
grep -E "plain|empty queue|inside the library" $O/host_profile_cfg1.txt
timeout -k 10 120 python bench.py --config cfg1 --steps 300 --warmup 20 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err || { tail -20 $O/cfg1.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/cfg1.json"))
print("cfg1", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "unblocked", round(d["config"]["host_enqueue_ms_per_step_unblocked"],3), "launches", d["config"]["launches_per_step"])
PY
timeout -k 10 200 python tools/epoch_bench.py 17118 32 cfg1 > $O/epoch_b32_cfg1.txt 2>&1 && grep epoch $O/epoch_b32_cfg1.txt
