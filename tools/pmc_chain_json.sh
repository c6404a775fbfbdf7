# FETCH_SIZE / WRITE_SIZE of the GRU level kernel alone (tools/chain_bench.py 1024 512: the launch BENCH.roofline times),
# one counter per run -> gpurun_out/prof/chain_pmc.json (bench.py: measured_chain_traffic reads profiles/rNN_chain_pmc.json)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_chain_$c -o p -- python3 $R/tools/chain_bench.py 1024 512 > $O/pmc_chain_$c.log 2>&1 || { tail -5 $O/pmc_chain_$c.log; exit 1; }
done
python3 - $O <<'PY'
import csv, collections, json, os, sys
O = sys.argv[1]
out = {"shape_MNK": [1024, 1536, 512], "kernel": "chain_kernel<3, false, 0, ...> (GRU level: h_prev W_hh^T + GRUCell tail)",
       "command": "rocprofv3 --pmc <counter> -- python3 tools/chain_bench.py 1024 512"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for r in csv.DictReader(open(os.path.join(O, "pmc_chain_" + c, "p_counter_collection.csv"))):
        if "chain_kernel<3, false, 0" in r["Kernel_Name"] and r["Counter_Name"] == c:
            v.append(float(r["Counter_Value"]))
    out[c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump(out, open(os.path.join(O, "chain_pmc.json"), "w"), indent=1)
print(out)
PY
