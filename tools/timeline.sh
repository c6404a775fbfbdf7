# usage: bash tools/timeline.sh <out.txt> <bench args...> : kernel timeline of the last step of a short bench run
R=${GRAFT_REPO_ROOT:-$PWD}
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/tl.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/tl/*kernel_trace.csv $R/gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/trace_timeline.py $f 6 -v > $R/$out 2>&1
rm -rf $R/gpurun_out/tl
