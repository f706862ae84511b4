#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the last training step's launch sequence."""
import csv, sys, collections
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(csv.DictReader(open(path)))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':80s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'us/step':>9s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{n[:80]:80s} {c:6d} {t/1e3:9.3f} {t/c:9.1f} {t/nsteps:9.1f}")
