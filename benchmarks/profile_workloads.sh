#!/bin/bash
# Per-workload profile artefacts on the GPU box (C3, C4, C5 by default; C2 has profile_round.sh):
#   gpurun_out/prof_bench_<w>.json          bench line of the workload (roofline + cpu_baseline)
#   gpurun_out/prof_<w>_kernel_stats.csv    rocprofv3 --kernel-trace --stats table of the single-stream run of the same command
#   gpurun_out/prof_<w>_streams1.json       bench line of that run
# usage: bash benchmarks/profile_workloads.sh [C3 C4 C5]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for W in ${@:-C3 C4 C5}; do
  w=$(echo $W | tr A-Z a-z)
  timeout -k 10 400 python3 bench.py --workload $W > $OUT/prof_bench_$w.json 2> $OUT/prof_bench_$w.err
  echo "$W bench done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$w -o kt -- python3 bench.py --workload $W --cpu-steps 0 --steps 30 --warmup 5 --roofline-steps 0 --streams 1 --no-extras > $OUT/prof_${w}_streams1.json 2> $OUT/prof_${w}_kt.err
  python3 - <<PY
import glob, shutil
st = glob.glob("$OUT/kt_$w/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(st, "$OUT/prof_${w}_kernel_stats.csv")
PY
  rm -rf $OUT/kt_$w
  echo "$W kernel trace done"
done
python3 - <<PY
import json
for w in "${@:-C3 C4 C5}".lower().split():
    d = json.loads(open(f"$OUT/prof_bench_{w}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(w, round(d["value"]), round(d["ms_per_step"], 3), "frac", round(r["frac"], 3), r["kernel"][:50], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
