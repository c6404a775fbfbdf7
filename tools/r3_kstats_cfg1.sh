#!/bin/bash
# per-kernel time of the batch-32 step, with and without the persistent small-batch chains
set -o pipefail
mkdir -p gpurun_out/r3a
DVAE_CHAIN_TINY_PERSIST=1 bash tools/kstats.sh cfg1_persist1 --config cfg1 --steps 100 --warmup 10 > gpurun_out/r3a/kstats_cfg1_persist1.txt 2>&1 || { tail gpurun_out/r3a/kstats_cfg1_persist1.txt; exit 1; }
DVAE_CHAIN_TINY_PERSIST=0 bash tools/kstats.sh cfg1_persist0 --config cfg1 --steps 100 --warmup 10 > gpurun_out/r3a/kstats_cfg1_persist0.txt 2>&1 || { tail gpurun_out/r3a/kstats_cfg1_persist0.txt; exit 1; }
python3 - <<'PY'
import csv
for tag in ("cfg1_persist1", "cfg1_persist0"):
    rows = list(csv.DictReader(open("gpurun_out/kstats/%s/s_kernel_stats.csv" % tag)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(tag, "total kernel ms per step", tot / 1e6 / 110)
    for r in rows[:40]:
        print("  %-90s %6s calls  avg %8.1f us  per step %7.1f us" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / 110))
PY
