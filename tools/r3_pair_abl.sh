#!/bin/bash
# ablation of pair_fwd_split_kernel (diagnostic builds of tools/pair_ablate.sh; time only): kernel durations from rocprofv3
set -o pipefail
O=gpurun_out/r3u
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 4 8 16 31; do
  lib=$R/exp_lib/libdvae_pabl$n.so
  [ $n = 0 ] && lib=$R/d-vae_amd/csrc/libdvae_hip.so
  DVAE_EXP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/abl$n -o s -- python3 $R/tools/bench_with_lib.py --config cfg5 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/abl$n.log 2>&1 || { tail $R/$O/abl$n.log; exit 1; }
  python3 - <<PY
import csv
for r in csv.DictReader(open("$R/$O/abl$n/s_kernel_stats.csv")):
    if "pair_fwd" in r["Name"]:
        print("PAIR_ABL=%-2s %-46s %3s calls  avg %8.1f us  min %8.1f" % ("$n", r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
