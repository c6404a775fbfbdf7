#!/bin/bash
set -o pipefail
for ab in 0 1; do
  DVAE_ACT_BF16=$ab bash tools/kstats.sh cfg5_ab$ab --config cfg5 --dtype bf16 --steps 3 --warmup 1 > gpurun_out/kstats_cfg5_ab$ab.txt 2>&1 || { tail gpurun_out/kstats_cfg5_ab$ab.txt; exit 1; }
  echo "== DVAE_ACT_BF16=$ab"; grep -E "gemm_bf16_nn|gru_fwd|gated_fwd|gather_sum" gpurun_out/kstats_cfg5_ab$ab.txt
done
