cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --steps 6 --warmup 3 > $R/gpurun_out/tl.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/tl/*kernel_trace.csv $R/gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/trace_timeline.py $f 6 -v > $R/gpurun_out/timeline.txt 2>&1
rm -rf $R/gpurun_out/tl
