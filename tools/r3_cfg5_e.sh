#!/bin/bash
set -o pipefail
for tb in 0 1; do
  DVAE_TABLES_BF16=$tb bash tools/kstats.sh cfg5_tb$tb --config cfg5 --dtype bf16 --steps 3 --warmup 1 > gpurun_out/kstats_cfg5_tb$tb.txt 2>&1 || { tail gpurun_out/kstats_cfg5_tb$tb.txt; exit 1; }
  echo "== DVAE_TABLES_BF16=$tb"; grep -E "pair_fwd|gemm_bf16_nn_kernel<256" gpurun_out/kstats_cfg5_tb$tb.txt
done
