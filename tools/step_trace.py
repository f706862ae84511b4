#!/usr/bin/env python3
"""Print the last training step of a rocprofv3 kernel trace (launch order, gaps, durations) and per-kernel totals."""
import csv, sys, collections
if sys.argv[1].endswith('.db'):                     # rocprofv3's default output (no --output-format csv)
    import sqlite3
    rows = [dict(Kernel_Name=n, Start_Timestamp=s, End_Timestamp=e) for n, s, e in
            sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels")]
else:
    rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw_kernel')]
a, b = idx[-3] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp']); tot = 0; prev = t0
agg = collections.OrderedDict()
verbose = len(sys.argv) > 2
for r in rows[a:b]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp']); d = (e - s) / 1e3; tot += d
    n = r['Kernel_Name'].split('(')[0][-60:]
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += d
    if verbose:
        print(f"{(s-t0)/1e3:9.1f} gap{(s-prev)/1e3:7.1f} {d:8.1f} {r['Kernel_Name'][:90]}")
    prev = e
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:62s} {c:4d} {t:9.1f}")
print('launches', b - a, 'kernel us', round(tot, 1), 'span us', (int(rows[b-1]['End_Timestamp']) - t0) / 1e3)
