#!/usr/bin/env python3
"""Per-queue, per-kernel totals of a tools/trace_timeline.py -v listing.  usage: python tools/timeline_agg.py <timeline.txt> [top]"""
import collections
import re
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
head = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) q(\d+) gap\s+([\d.]+) dur\s+([\d.]+) wg\s+(\d+) (.*)', l)
    if not m:
        head.append(l.rstrip())
        continue
    key = (m.group(2), m.group(6)[:80])
    agg[key][0] += 1
    agg[key][1] += float(m.group(4))
print("\n".join(head[:6]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for (q, name), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("q%s %-82s %5d  %9.3f ms  avg %8.1f us" % (q, name, n, t / 1e3, t / n))
