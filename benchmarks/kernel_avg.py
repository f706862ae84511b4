#!/usr/bin/env python3
"""Average duration (us) of the kernels whose name contains argv[2] in a rocprofv3 results .db (argv[1])."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
r = [(e - s) / 1e3 for n, s, e in c.execute("select name, start, end from kernels") if sys.argv[2] in n]
skip = min(3, len(r) // 2)
print(sys.argv[2], "n", len(r), "avg us %.1f" % (sum(r[skip:]) / max(len(r) - skip, 1)))
