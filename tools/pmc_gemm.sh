cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "4096 4096 4096 0 0 0" "7168 2048 512 0 0 0"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_$tag -o p --output-format csv -- python3 $R/tools/gemm_one.py $shape > $R/gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
