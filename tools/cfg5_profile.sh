# Kernel stats + launch log of the cfg5 bf16 step (run on the GPU box through gpurun) -> gpurun_out/cfg5/
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/cfg5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --config cfg5 --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1 || exit 1
cd $R
DVAE_GEMM_LOG=1 timeout -k 10 300 python bench.py --config cfg5 --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline > $O/log.json 2> $O/launch_log.txt || exit 1
