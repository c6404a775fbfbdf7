# Collects the round's profile artifacts into gpurun_out/prof/ (run on the GPU box through gpurun).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel stats of the bench workload, both matrix-unit modes
for dt in f32 bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$dt -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype $dt > $O/stats_$dt.log 2>&1 || exit 1
done
# 2. PMC passes on the dominant GEMM launch, one counter group per run
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/gemm_one.py 9251 2048 512 0 0 0 > $O/pmc_$tag.log 2>&1 || exit 1
done
cd $R
# 3. plain runs
timeout -k 10 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 1
timeout -k 10 300 python bench.py --dtype bf16 > $O/bench_cfg2_bf16.json 2> $O/bench_cfg2_bf16.err || exit 1
timeout -k 10 100 python tools/gemm_bench.py > $O/gemm_shapes_fp32.txt 2>&1 || exit 1
DVAE_GEMM_BF16=1 timeout -k 10 100 python tools/gemm_bench.py > $O/gemm_shapes_bf16.txt 2>&1 || exit 1
timeout -k 10 100 python tools/phase_times.py > $O/phase_times_cfg2.txt 2>&1 || exit 1
timeout -k 10 200 python tools/roofline_by_kernel.py --log $O/launch_log_cfg2.txt > $O/launch_log.out 2>&1 || exit 1
timeout -k 10 200 python tools/phase_times.py cfg5 4 bf16 > $O/phase_times_cfg5_bf16.txt 2>&1 || exit 1
timeout -k 10 200 python tools/host_cost.py cfg1 10 > $O/host_cost_cfg1.txt 2>&1 || exit 1
for st in f32xf32 f32xbf16 bf16xbf16; do
  DVAE_GEMM_BF16=1 DVAE_GEMM_STORE=$st timeout -k 10 100 python tools/gemm_bench.py > $O/gemm_shapes_bf16_$st.txt 2>&1 || exit 1
done
# 4. cfg 5 in bf16 mode (round 2: large weight gradients, wide pair tables, bf16 level kernels)
timeout -k 10 200 python tools/wgrad_bench.py > $O/wgrad_bf16_256.txt 2>&1 || exit 1
DVAE_WGRAD_256=0 timeout -k 10 200 python tools/wgrad_bench.py > $O/wgrad_bf16_128.txt 2>&1 || exit 1
timeout -k 10 200 python tools/chain_bf16_bench.py > $O/chain_bf16_levels.txt 2>&1 || exit 1
bash tools/kstats.sh cfg5_bf16 --config cfg5 --dtype bf16 --steps 3 --warmup 1 > $O/kstats_cfg5_bf16.txt 2>&1 || exit 1
cp $R/gpurun_out/kstats/cfg5_bf16/s_kernel_stats.csv $O/cfg5_bf16_kernel_stats.csv || exit 1
# 5. whole epochs including shuffling and packing (tools/epoch_bench.py)
timeout -k 10 200 python tools/epoch_bench.py 17118 32 cfg1 > $O/epoch_b32_cfg1.txt 2>&1 || exit 1
timeout -k 10 200 python tools/epoch_bench.py 17118 32 cfg2 > $O/epoch_b32_cfg2.txt 2>&1 || exit 1
timeout -k 10 200 python tools/epoch_bench.py 17118 1024 cfg2 > $O/epoch_b1024_cfg2.txt 2>&1 || exit 1
# 6. round 3: the batch-32 path (small-M product kernel, persistent chains, host side)
timeout -k 10 200 python tools/gemm_skinny_bench.py > $O/gemm_skinny.txt 2>&1 || exit 1
timeout -k 10 200 python tools/host_profile.py cfg1 300 > $O/host_profile_cfg1.txt 2>&1 || exit 1
bash tools/timeline.sh gpurun_out/prof/timeline_cfg1.txt --config cfg1 --steps 10 --warmup 5 || exit 1
bash tools/timeline.sh gpurun_out/prof/timeline_cfg5_bf16.txt --config cfg5 --dtype bf16 --steps 3 --warmup 2 || exit 1
python3 tools/timeline_agg.py $O/timeline_cfg5_bf16.txt 45 > $O/timeline_cfg5_bf16_by_kernel.txt || exit 1
bash tools/kstats.sh cfg1 --config cfg1 --steps 100 --warmup 10 > $O/kstats_cfg1.txt 2>&1 || exit 1
cp $R/gpurun_out/kstats/cfg1/s_kernel_stats.csv $O/cfg1_kernel_stats.csv || exit 1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain_alone -o s -- python3 $R/tools/chain_bench.py 1024 512 > $O/chain_alone.log 2>&1 || exit 1
cp $O/chain_alone/s_kernel_stats.csv $O/chain_level_alone_kernel_stats.csv || exit 1
cd $R
# 7. PMC traffic of the GRU level kernel alone (BENCH.roofline.traffic)
bash tools/pmc_chain_json.sh > $O/pmc_chain_json.log 2>&1 || exit 1
