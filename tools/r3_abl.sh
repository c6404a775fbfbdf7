#!/bin/bash
# ablation of the persistent small-batch chains (diagnostic builds; time only): kernel durations from rocprofv3
set -o pipefail
O=gpurun_out/r3h
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 0 1 5 7; do
  lib=$R/exp_lib/libdvae_tabl$n.so
  [ $n = 0 ] && lib=$R/d-vae_amd/csrc/libdvae_hip.so
  DVAE_EXP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/abl$n -o s -- python3 $R/tools/bench_with_lib.py --config cfg1 --steps 60 --warmup 10 --no-cpu-baseline > $R/$O/abl$n.log 2>&1 || { tail $R/$O/abl$n.log; exit 1; }
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/$O/abl$n/s_kernel_stats.csv")))
for r in rows:
    if "tiny_chain" in r["Name"]:
        print("TINY_ABL=$n  %-50s %5s calls  avg %7.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
