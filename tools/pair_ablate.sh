# Diagnostic variants of the wide pair forward (csrc/loss_kernels.hip: PAIR_ABL) into exp_lib/ (CPU container), timed on
# the GPU with tools/r3_pair_abl.sh.  usage: bash tools/pair_ablate.sh 1 2 4 8 16 ...
cd "$(dirname "$0")/../d-vae_amd/csrc" || exit 1
mkdir -p ../../exp_lib
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DPAIR_ABL=$n -c loss_kernels.hip -o /tmp/loss_kernels_abl$n.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm.o gemm_skinny.o gemm_bf16.o gemm_bf16_nn.o chain.o chain_persist.o pointwise.o /tmp/loss_kernels_abl$n.o pack.o optim.o train.o decode.o api.o packer.o -o ../../exp_lib/libdvae_pabl$n.so || exit 1
done
