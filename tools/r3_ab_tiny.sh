#!/bin/bash
# same-box A/B of the persistent small-batch chains: HEAD's kernels (exp_lib/libdvae_base.so) against the working tree's
set -o pipefail
O=gpurun_out/r3k
mkdir -p $O
R=$PWD
for tag in base cur; do
  lib=$R/exp_lib/libdvae_base.so; [ $tag = cur ] && lib=$R/d-vae_amd/csrc/libdvae_hip.so
  DVAE_EXP_LIB=$lib timeout -k 10 120 python tools/host_cost.py cfg1 20 > $O/host_cost_$tag.txt 2>&1 || { tail $O/host_cost_$tag.txt; exit 1; }
  echo "== $tag"; grep "host enqueue" $O/host_cost_$tag.txt
done
cd /tmp && export TMPDIR=/tmp
for tag in base cur; do
  lib=$R/exp_lib/libdvae_base.so; [ $tag = cur ] && lib=$R/d-vae_amd/csrc/libdvae_hip.so
  DVAE_EXP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$tag -o s -- python3 $R/tools/bench_with_lib.py --config cfg1 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/$tag.log 2>&1 || { tail $R/$O/$tag.log; exit 1; }
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/$O/$tag/s_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("$tag total kernel ms per step", tot / 1e6 / 110)
for r in rows[:45]:
    print("  %-100s %6s calls  avg %8.1f us  per step %7.1f us" % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / 110))
PY
done > $R/$O/kstats.txt
grep -E "total|tiny_chain" $R/$O/kstats.txt
cd $R
DVAE_GEMM_LOG=1 timeout -k 10 120 python bench.py --config cfg1 --steps 1 --warmup 0 --no-cpu-baseline > $O/gemmlog.json 2> $O/gemmlog.txt || { tail $O/gemmlog.txt; exit 1; }
grep -c GEMMSEL $O/gemmlog.txt
