#!/bin/bash
# Average duration of our kernels in a short single-stream bench run under rocprofv3 --kernel-trace (second half of
# the launches of each kernel), optional filter as $1.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/kt1 -o r -- python3 bench.py --cpu-steps 0 --steps 6 --warmup 3 --roofline-steps 0 --streams 1 --no-extras $BENCH_EXTRA > /dev/null 2>&1
python3 - "$1" <<PY
import sqlite3, glob, collections, sys
db = glob.glob("$ROOT/gpurun_out/kt1/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
r = collections.defaultdict(list)
for n, s, e in c.execute("select name, start, end from kernels"):
    if "at::" in n or "rocprim" in n or "rocclr" in n: continue
    r[n].append((e - s) / 1e3)
tot = 0
for k, v in sorted(r.items(), key=lambda kv: -sum(kv[1])):
    h = v[len(v) // 2:]; a = sum(h) / len(h)
    per_step = a * (len(v) / 9.0)
    if "cast_bf16" in k: continue
    tot += per_step
    if not sys.argv[1] or sys.argv[1] in k: print("%-64s avg %8.1f us  x %4.1f per step" % (k.split("(")[0][-64:], a, len(v) / 9.0))
print("per step total %.1f us" % tot)
PY
rm -rf $ROOT/gpurun_out/kt1
