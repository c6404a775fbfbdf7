cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base bands; do
  if [ $v = bands ]; then export DVAE_EXP_LIB=$R/exp_lib/lib_bands.so; else unset DVAE_EXP_LIB; fi
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcab_$v -o p -- python3 $R/tools/gemm_one.py 9251 2048 512 0 0 0 > $R/gpurun_out/pmcab_$v.log 2>&1 || exit 1
done
cd $R
for i in 1 2; do echo "== base"; timeout -k 10 100 python tools/gemm_bench.py || exit 1; echo "== bands"; DVAE_EXP_LIB=exp_lib/lib_bands.so timeout -k 10 100 python tools/gemm_bench.py || exit 1; done > gpurun_out/ab.txt 2>&1
