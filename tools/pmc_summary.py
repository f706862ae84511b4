#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-50:]
    if filt and filt not in r['Kernel_Name']:
        continue
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[n]['_dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    agg[n]['_vgpr'].append(float(r['VGPR_Count'])); agg[n]['_lds'].append(float(r['LDS_Block_Size']))
for n, d in agg.items():
    print(n)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
