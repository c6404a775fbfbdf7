# Builds diagnostic variants of the library with parts of the persistent small-batch chains removed (results wrong,
# time only) into exp_lib/ (CPU container), to be timed on the GPU with
#   DVAE_EXP_LIB=exp_lib/libdvae_tabl<N>.so python tools/bench_with_lib.py --config cfg1 ...
# TINY_ABL bits: 1 no waits, 2 plain loads of the handed-off operand, 4 no drain + signal.
cd "$(dirname "$0")/../d-vae_amd/csrc" || exit 1
mkdir -p ../../exp_lib
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DTINY_ABL=$n -c chain_persist.hip -o /tmp/chain_persist_abl$n.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm.o gemm_bf16.o gemm_bf16_nn.o chain.o /tmp/chain_persist_abl$n.o pointwise.o loss_kernels.o pack.o optim.o train.o decode.o api.o packer.o -o ../../exp_lib/libdvae_tabl$n.so || exit 1
done
# stage timing build: TINY_PROF=1 -> exp_lib/libdvae_tprof.so (tools/tiny_prof.py reads the stage sums)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -DTINY_PROF=1 -c chain_persist.hip -o /tmp/chain_persist_prof.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm.o gemm_bf16.o gemm_bf16_nn.o chain.o /tmp/chain_persist_prof.o pointwise.o loss_kernels.o pack.o optim.o train.o decode.o api.o packer.o -o ../../exp_lib/libdvae_tprof.so || exit 1
