"""Launches of the last paired step in a rocprofv3 --kernel-trace csv (steps are delimited by the adamw_kernel launches,
two per paired step): python benchmarks/step_summary.py <dir with *kernel_trace.csv> <title>"""
import csv, glob, collections, sys
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
a, b = idx[-3] + 1, idx[-1] + 1
agg = collections.OrderedDict(); tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    n = r["Kernel_Name"].split("(")[0][-72:]
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += d
print(sys.argv[2] if len(sys.argv) > 2 else tr, " (last paired step: kernel, launches, us)")
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:74s} {k:4d} {t:9.1f}")
print(f"launches {b - a} kernel us {tot:.1f} span us {(int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f}")
