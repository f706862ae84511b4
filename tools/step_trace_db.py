#!/usr/bin/env python3
"""step_trace.py for a rocprofv3 rocpd database (kernels view): last training step, per-kernel totals."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels order by start")]
idx = [i for i, r in enumerate(rows) if r['name'].startswith('adamw_kernel')]
a, b = idx[-3] + 1, idx[-1] + 1
agg = collections.OrderedDict(); tot = 0
for r in rows[a:b]:
    d = (r['end'] - r['start']) / 1e3; tot += d
    n = r['name'].split('(')[0][-70:]
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += d
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:72s} {k:4d} {t:9.1f}")
print('launches', b - a, 'kernel us', round(tot, 1), 'span us', (rows[b-1]['end'] - rows[a]['start']) / 1e3)
