# all configurations x both matrix-unit modes, one JSON line each -> gpurun_out/bench_all.jsonl
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/bench_all.jsonl
: > $out
for cfg in cfg1 cfg3 cfg4 cfg5; do
  for dt in f32 bf16; do
    steps=20; [ $cfg = cfg5 ] && steps=4; [ $cfg = cfg1 ] && steps=300  # (a batch-32 step is 1 ms: 20 of them are mostly queue depth)
    timeout -k 10 300 python $R/bench.py --config $cfg --dtype $dt --steps $steps --warmup 2 --no-cpu-baseline >> $out 2>> $R/gpurun_out/bench_all.err || exit 1
  done
done
