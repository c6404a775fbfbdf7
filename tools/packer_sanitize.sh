#!/bin/bash
# Host-code sanitizer run of the DAG packer (AddressSanitizer + UBSan on the HOST side only:
# -fsanitize directly after -Xarch_host): 300 random batches (all shapes up to max_n = 64,
# nvt = 32), every 7th one malformed, through dvae_batch_create / export / destroy.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/d-vae_amd/csrc
T=$(mktemp -d)
F="-O1 -g -std=c++17 -x hip --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/packer.cpp -o $T/packer.o
/opt/rocm/bin/hipcc $F -c $C/api.cpp -o $T/api.o
/opt/rocm/bin/hipcc $F -c $R/tools/packer_sanitize_driver.cpp -o $T/drv.o
/opt/rocm/bin/hipcc -fsanitize=address,undefined --offload-arch=gfx950 $T/packer.o $T/api.o $T/drv.o -o $T/drv
ASAN_OPTIONS=detect_leaks=1 $T/drv
rm -rf $T
