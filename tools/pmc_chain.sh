# PMC passes on the chain level kernels (tools/chain_bench.py), one counter group per run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_chain
mkdir -p $O
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/p$i -o p -- python3 $R/tools/chain_bench.py > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; exit 1; }
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain_kernel" in k or "gemm_f32" in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s mean %14.1f  n %d" % (c, sum(v) / len(v), len(v)))
PY
