#!/bin/bash
# the reference's ENAS recipe (README.md:37: 300 epochs, batch 32) on the reference's real rows, end to end on the GPU path
set -o pipefail
mkdir -p gpurun_out/r3e
timeout -k 10 1000 python -u tools/train_demo.py 300 0 32 --data final_structures6 2>&1 | tee gpurun_out/r3e/train_real_enas6_300_epochs.txt
