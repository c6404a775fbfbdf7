#!/bin/bash
# round 3: full GPU suite after the robustness changes, bench lines (new fields), then the real-data 300-epoch run
set -o pipefail
O=gpurun_out/r3d
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 200 python bench.py --steps 50 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err || { tail -20 $O/bench_cfg2.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_cfg2.json"))
print("cfg2", round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms; roofline", d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],3), "gemm", round(d["roofline_gemm"]["frac"],3), "host unblocked", round(d["config"]["host_enqueue_ms_per_step_unblocked"],3), "host", round(d["config"]["host_enqueue_ms_per_step"],3))
PY
DVAE_FORCE_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 200 python tools/bench_force_collective.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_collective_overlap.json 2> $O/bench_cfg2_collective_overlap.err || { tail -20 $O/bench_cfg2_collective_overlap.err; exit 1; }
DVAE_FORCE_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout -k 10 200 python tools/bench_force_collective.py --steps 30 --warmup 5 --no-cpu-baseline --allreduce after > $O/bench_cfg2_collective_after.json 2> $O/bench_cfg2_collective_after.err || { tail -20 $O/bench_cfg2_collective_after.err; exit 1; }
python - <<PY
import json
for m in ("overlap","after"):
    d=json.load(open("$O/bench_cfg2_collective_%s.json" % m))
    c=d["config"]
    print(m, round(d["value"]), "DAGs/s; allreduce exposed", round(c["allreduce_ms_per_step"],4), "total", round(c["allreduce_total_ms"],4), "early MB", round(c["allreduce_early_mb"],2), "enc bwd", c["encode_bwd_ms"])
PY
