"""Copies the artifacts of tools/profile_round.sh (+ tools/bench_all.sh) from gpurun_out/ into
profiles/ under the round's names.  usage: python tools/install_profiles.py r01"""
import csv
import collections
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
D = os.path.join(ROOT, "profiles")
r = sys.argv[1] if len(sys.argv) > 1 else "r01"


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


for src, dst in [("bench_cfg2.json", "bench_cfg2.json"), ("bench_cfg2_bf16.json", "bench_cfg2_bf16.json")]:
    json.dump(last_json(os.path.join(P, src)), open(os.path.join(D, "%s_%s" % (r, dst)), "w"))
shutil.copy(os.path.join(P, "stats_f32", "s_kernel_stats.csv"), os.path.join(D, "%s_bench_cfg2_kernel_stats.csv" % r))
shutil.copy(os.path.join(P, "stats_bf16", "s_kernel_stats.csv"), os.path.join(D, "%s_bench_cfg2_bf16_kernel_stats.csv" % r))
for f in ("gemm_shapes_fp32.txt", "gemm_shapes_bf16.txt", "phase_times_cfg2.txt", "launch_log_cfg2.txt",
          "phase_times_cfg5_bf16.txt", "host_cost_cfg1.txt", "gemm_shapes_bf16_f32xf32.txt", "gemm_shapes_bf16_f32xbf16.txt",
          "gemm_shapes_bf16_bf16xbf16.txt", "epoch_b32_cfg1.txt", "epoch_b32_cfg2.txt", "epoch_b1024_cfg2.txt",
          "wgrad_bf16_256.txt", "wgrad_bf16_128.txt", "cfg5_bf16_kernel_stats.csv", "gemm_skinny.txt",
          "host_profile_cfg1.txt", "timeline_cfg1.txt", "timeline_cfg5_bf16_by_kernel.txt", "cfg1_kernel_stats.csv", "chain_level_alone_kernel_stats.csv"):
    if not os.path.exists(os.path.join(P, f)):
        continue
    txt = "".join(l for l in open(os.path.join(P, f)) if "amdgpu.ids" not in l)
    open(os.path.join(D, "%s_%s" % (r, f)), "w").write(txt)

pm = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(os.path.join(P, "pmc_" + tag, "p_counter_collection.csv"))):
        if "gemm_f32" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        pm[k] = (sum(v) / len(v), len(v))
print("PMC:", pm)
pmj = {k: {"mean": v[0], "launches": v[1]} for k, v in pm.items()}
pmj["shape_MNK"] = [9251, 2048, 512]  # tools/profile_round.sh: gemm_one.py 9251 2048 512 (the EB product of cfg 2)
json.dump(pmj, open(os.path.join(D, "%s_gemm_pmc.json" % r), "w"), indent=1)

if os.path.exists(os.path.join(P, "chain_pmc.json")):  # tools/pmc_chain_json.sh: the GRU level kernel alone
    shutil.copy(os.path.join(P, "chain_pmc.json"), os.path.join(D, "%s_chain_pmc.json" % r))

allp = os.path.join(ROOT, "gpurun_out", "bench_all.jsonl")
if os.path.exists(allp):
    out = {}
    for l in open(allp):
        if l.startswith("{"):
            v = json.loads(l)
            key = v["config"]["workload"].split(":")[0] + ("_bf16" if v["dtype"] == "bf16" else "")
            out[key] = v
    json.dump(out, open(os.path.join(D, "%s_bench_other_configs.json" % r), "w"), indent=1)
    print("other configs:", {k: round(v["value"]) for k, v in out.items()})
