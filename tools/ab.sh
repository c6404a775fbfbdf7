# usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   : one bench line per environment, interleaved twice
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2; do
for e in "$@"; do
  out=$(env $e timeout -k 10 200 python tools/bench_with_lib.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %8.0f DAGs/s  %.3f ms' % ('$e', d['value'], d['ms_per_step']))" || exit 1
done
done
