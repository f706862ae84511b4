#!/bin/bash
# Produces the per-round profile artefacts on the GPU box (small summaries only; raw traces are deleted):
#   gpurun_out/prof_bench.json            default bench line (roofline + cpu_baseline)
#   gpurun_out/prof_streams1.json         bench line of the single-stream run under rocprofv3 --kernel-trace
#   gpurun_out/prof_kernel_stats.csv      rocprofv3 --stats per-kernel table of that run
#   gpurun_out/prof_step_summary.txt      launches of the last paired step of that run
#   gpurun_out/prof_sq_counters.txt       SQ counters per kernel (separate --pmc pass)
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/prof_bench.json 2> $OUT/prof_bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --cpu-steps 0 --steps 90 --warmup 5 --roofline-steps 0 --streams 1 --no-extras > $OUT/prof_streams1.json 2> $OUT/prof_kt.err
echo "kernel trace done"
python3 - <<PY
import csv, glob, collections, shutil
st = glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(st, "$OUT/prof_kernel_stats.csv")
tr = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
a, b = idx[-3] + 1, idx[-1] + 1
agg = collections.OrderedDict(); tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    n = r["Kernel_Name"].split("(")[0][-64:]
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += d
with open("$OUT/prof_step_summary.txt", "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-steps 0 --steps 90 --warmup 5 --roofline-steps 0 --streams 1  (last paired step: kernel, launches, us)\n")
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write(f"{n:66s} {k:4d} {t:9.1f}\n")
    fh.write(f"launches {b - a} kernel us {tot:.1f} span us {(int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f}\n")
PY
rm -rf $OUT/kt
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
timeout -s KILL 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc -o pmc -- python3 bench.py --cpu-steps 0 --steps 4 --warmup 2 --roofline-steps 0 --streams 1 --no-extras > /dev/null 2> $OUT/prof_pmc.err
echo "pmc done"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
cnt = "$CNT".split()
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    full = r["Kernel_Name"]
    if "at::" in full or "rocprim" in full or "rocclr" in full:
        continue
    n = full.split("(")[0][-46:]
    agg.setdefault(n, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/prof_sq_counters.txt", "w") as fh:
    fh.write("rocprofv3 --pmc " + " ".join(cnt) + " -- python3 bench.py --cpu-steps 0 --steps 4 --warmup 2 --roofline-steps 0 --streams 1  (means of the last two launches)\n")
    fh.write(f"{'kernel':46s} " + " ".join(f"{c[3:16]:>14s}" for c in cnt) + "  mfma_busy  valu/wave\n")
    for n, d in agg.items():
        m = {c: (sum(d[c][-2:]) / max(len(d[c][-2:]), 1) if d[c] else 0.0) for c in cnt}
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_BUSY_CU_CYCLES"]) if m["SQ_BUSY_CU_CYCLES"] else 0.0
        vw = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
        fh.write(f"{n:46s} " + " ".join(f"{m[c]:14.3g}" for c in cnt) + f" {busy:10.3f} {vw:10.3f}\n")
PY
rm -rf $OUT/pmc
# second counter pass: LDS activity of every kernel + the decoders' retired vector instructions -> prof_decoder_valu.json
# (what bench.py's roofline.decoders prices the VALU-issue bound with)
CNT2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32"
timeout -s KILL 240 rocprofv3 --pmc $CNT2 --output-format csv -d $OUT/pmc2 -o pmc -- python3 bench.py --cpu-steps 0 --steps 4 --warmup 2 --roofline-steps 0 --streams 1 --no-extras > /dev/null 2> $OUT/prof_pmc2.err
echo "pmc2 done"
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/pmc2/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    full = r["Kernel_Name"]
    if "at::" in full or "rocprim" in full or "rocclr" in full:
        continue
    agg.setdefault(full.split("(")[0], collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(d, c):
    v = d.get(c, [])[-2:]
    return sum(v) / len(v) if v else 0.0
with open("$OUT/prof_lds_counters.txt", "w") as fh:
    fh.write("rocprofv3 --pmc $CNT2 -- python3 bench.py --cpu-steps 0 --steps 4 --warmup 2 --roofline-steps 0 --streams 1 --no-extras  (means of the last two launches)\n")
    fh.write(f"{'kernel':50s} {'INSTS_VALU':>12s} {'VALU_TRANS':>12s} {'INSTS_LDS':>12s} {'LDS_IDX_ACT':>12s} {'BANK_CONFL':>12s} {'conflict share':>15s}\n")
    for n, d in agg.items():
        act, bc = mean(d, "SQ_LDS_IDX_ACTIVE"), mean(d, "SQ_LDS_BANK_CONFLICT")
        fh.write(f"{n[-50:]:50s} {mean(d, 'SQ_INSTS_VALU'):12.4g} {mean(d, 'SQ_INSTS_VALU_TRANS_F32'):12.4g} {mean(d, 'SQ_INSTS_LDS'):12.4g} {act:12.4g} {bc:12.4g} {(bc / act if act else 0.0):15.3f}\n")
B, G, P = 1024, 20000, 100000
kern = {"ber": {"names": ["dec_kernel<1, 48, 2, 0, true>"], "elements": B * P, "trans_src": 4.0},
        "nb": {"names": ["dec_kernel<1, 48, 4, 2, true>", "dec_kernel<1, 48, 4, 3, true>", "dec_kernel<1, 48, 2, 4, true>"], "elements": B * G,
               "trans_src": 2.0 + 5.3 + 6.3}}
out = {"workload": "C2", "batch": B, "genes": G, "peaks": P, "command": "rocprofv3 --pmc $CNT2 -- python3 bench.py --cpu-steps 0 --steps 4 --warmup 2 --roofline-steps 0 --streams 1 --no-extras",
       "note": "insts_valu = SQ_INSTS_VALU per step summed over the entry point's dec_kernel passes (wave64 instructions, whole chip); transcendentals_per_element = SQ_INSTS_VALU_TRANS_F32 x 64 / label elements when that counter reports, else counted in the source (exp, log, rcp per element, summed over the passes)", "kernels": {}}
for k, spec in kern.items():
    iv = tr = act = bc = 0.0
    for n, d in agg.items():
        if any(n.endswith(x) for x in spec["names"]):
            iv += mean(d, "SQ_INSTS_VALU"); tr += mean(d, "SQ_INSTS_VALU_TRANS_F32"); act += mean(d, "SQ_LDS_IDX_ACTIVE"); bc += mean(d, "SQ_LDS_BANK_CONFLICT")
    if iv:
        out["kernels"][k] = {"insts_valu": iv, "transcendentals_per_element": (tr * 64.0 / spec["elements"]) if tr else spec["trans_src"],
                             "transcendentals_from": "SQ_INSTS_VALU_TRANS_F32" if tr else "source count",
                             "lds_bank_conflict_share": (bc / act if act else None)}
json.dump(out, open("$OUT/prof_decoder_valu.json", "w"), indent=1)
PY
rm -rf $OUT/pmc2
echo "all done"
