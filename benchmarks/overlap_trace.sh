#!/bin/bash
# Kernel timeline of the default two-stream bench step (rocprofv3 --kernel-trace): start / end / queue of every launch of one
# step, the union of busy time and the sum of durations -- how much the two direction streams actually overlap.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ov -o ov -- python3 bench.py --cpu-steps 0 --steps 30 --warmup 5 --roofline-steps 0 --no-extras $BENCH_EXTRA > /dev/null 2>&1
python3 - <<PY
import csv, glob
tr = glob.glob("$ROOT/gpurun_out/ov/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "at::" not in r["Kernel_Name"] and "rocprim" not in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void spmm_dw128_bf16_kernel<false")]
a, b = idx[-12], idx[-11]          # one step somewhere in the timed region: from one ATAC K2' launch to the next
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"])
with open("$ROOT/gpurun_out/overlap_step.txt", "w") as fh:
    tot = 0
    ev = []
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        tot += e - s
        ev.append((s, e))
        fh.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'].split('(')[0][-60:]}\n")
    ev.sort()
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(e for _, e in ev) - min(s for s, _ in ev)
    fh.write(f"launches {len(seg)}  sum of durations {tot/1e3:.1f} us  union busy {busy/1e3:.1f} us  span {span/1e3:.1f} us\n")
PY
rm -rf $ROOT/gpurun_out/ov
