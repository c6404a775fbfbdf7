# usage: tools/kstats.sh <tag> <bench args...> : rocprofv3 kernel stats of a short bench run -> gpurun_out/kstats/<tag>/
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
O=$R/gpurun_out/kstats/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/run.log 2>&1 || exit 1
python3 - $O/s_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s %6s calls  avg %10.1f us  total %9.3f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
