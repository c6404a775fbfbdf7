#!/bin/bash
# round 3: tail-wave signal; cfg 1 bench + timeline of one batch-32 step
set -o pipefail
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or random_graphs or edge_cases or n64" --timeout 120 --timeout-method thread -p no:cacheprovider > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -2 $O/parity.log
DVAE_CHAIN_TINY_PERSIST=1 timeout -k 10 120 python bench.py --config cfg1 --steps 200 --warmup 20 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err || { tail -20 $O/cfg1.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/cfg1.json"))
print(round(d["value"]), "DAGs/s", round(d["ms_per_step"],4), "ms/step host", round(d["config"]["host_enqueue_ms_per_step"],3), "launches", d["config"]["launches_per_step"])
PY
timeout -k 10 120 python tools/host_cost.py cfg1 20 > $O/host_cost.txt 2>&1 && cat $O/host_cost.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --config cfg1 --steps 6 --warmup 3 --no-cpu-baseline > $R/$O/tl.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/tl/*kernel_trace.csv $R/gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/trace_timeline.py $f 6 -v > $R/$O/timeline_cfg1.txt 2>&1
rm -rf $R/gpurun_out/tl
head -8 $R/$O/timeline_cfg1.txt
