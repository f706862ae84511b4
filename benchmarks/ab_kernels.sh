#!/bin/bash
# A/B of kernel durations: runs the same bench command under rocprofv3 --kernel-trace from _old/ (a copy of an earlier
# tree, built) and from the working tree, and prints the average duration of every kernel side by side.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
ARGS="--cpu-steps 0 --steps 12 --warmup 3 --roofline-steps 0 --streams 1 --no-extras $BENCH_EXTRA"
(cd _old && timeout -k 10 150 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/ab_old -o r -- python3 bench.py $ARGS > /dev/null 2>&1)
timeout -k 10 150 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/ab_new -o r -- python3 bench.py $ARGS > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob, collections
def load(d):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    r = collections.defaultdict(list)
    for n, s, e in c.execute("select name, start, end from kernels"):
        r[n].append((e - s) / 1e3)
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in r.items()}, {k: len(v) for k, v in r.items()}
a, na = load("$ROOT/gpurun_out/ab_old"); b, nb = load("$ROOT/gpurun_out/ab_new")
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k) or a.get(k))):
    if (b.get(k) or a.get(k)) > 3 and "at::" not in k and "rocprim" not in k: print("%-70s old %8.1f (%3d)  new %8.1f (%3d)" % (k[:70], a.get(k, 0), na.get(k, 0), b.get(k, 0), nb.get(k, 0)))
PY
rm -rf $ROOT/gpurun_out/ab_old $ROOT/gpurun_out/ab_new
