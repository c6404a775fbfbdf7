# Builds diagnostic variants of the library with parts of the chain level kernels removed (results wrong, time only)
# into exp_lib/ (CPU container), to be timed on the GPU with:  DVAE_EXP_LIB=exp_lib/libdvae_abl<N>.so python tools/chain_bench.py
cd "$(dirname "$0")/../d-vae_amd/csrc" || exit 1
mkdir -p ../../exp_lib
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DCHAIN_ABL=$n -c chain.hip -o /tmp/chain_abl$n.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm.o gemm_bf16.o gemm_bf16_nn.o /tmp/chain_abl$n.o pointwise.o loss_kernels.o pack.o optim.o train.o decode.o api.o packer.o -o ../../exp_lib/libdvae_abl$n.so || exit 1
done
