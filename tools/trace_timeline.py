#!/usr/bin/env python3
"""Timeline of one train step from a `rocprofv3 --kernel-trace --output-format csv` file:
per-queue busy time, the union of busy intervals, and idle gaps on the critical queue.
usage: python tools/trace_timeline.py <kernel_trace.csv> [n_steps_in_trace]"""
import csv
import re
import sys


def short(name):
    name = name.split("(")[0] if not name.startswith("void at::") else name
    name = name.replace("void ", "").replace("dvae::", "")
    return name[:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
           int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // 256) for r in rows]
    ks.sort()
    # steps are delimited by the fused Adam kernel
    ends = [i for i, k in enumerate(ks) if "FusedOptimizer" in k[3] or "multi_tensor_apply_kernel<at::native::(anonymous namespace)::FusedOpt" in k[3]]
    adam = [i for i, k in enumerate(ks) if "multi_tensor_apply" in k[3] or k[3].startswith("adam_kernel")]
    print("kernels", len(ks), "adam launches at", adam[-8:])
    if len(adam) < 2:
        return
    # the last step = kernels between the second-to-last and the last Adam launch
    a0 = adam[-2] if ks[adam[-1]][0] - ks[adam[-2]][0] > 1e6 else adam[-3]
    lo, hi = a0 + 1, adam[-1] + 1
    step = ks[lo:hi]
    t0 = step[0][0]
    t1 = max(k[1] for k in step)
    print("step: %d kernels, span %.1f us" % (len(step), (t1 - t0) / 1e3))
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, q, n, g in step:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("GPU busy (union) %.1f us, idle %.1f us" % (busy / 1e3, (t1 - t0 - busy) / 1e3))
    byq = {}
    for s, e, q, n, g in step:
        byq.setdefault(q, []).append((s, e, n))
    for q, v in sorted(byq.items()):
        print("queue %d: %d kernels, sum %.1f us" % (q, len(v), sum(e - s for s, e, _ in v) / 1e3))
    if "-v" in sys.argv:
        prev_e = {}
        for s, e, q, n, g in step:
            gap = (s - prev_e[q]) / 1e3 if q in prev_e else 0.0
            print("%9.1f q%d gap %6.1f dur %7.1f wg %6d %s" % ((s - t0) / 1e3, q, gap, (e - s) / 1e3, g, n))
            prev_e[q] = e


main()
