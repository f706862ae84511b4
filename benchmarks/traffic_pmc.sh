#!/bin/bash
# HBM bytes of the K2' launches from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (kernel-trace / stats only), unit KiB, reads = 2 x FETCH_SIZE on gfx950 for 16-byte-per-lane
# streaming loads.  Writes gpurun_out/prof_c2_traffic.json (copied to profiles/r03_c2_traffic.json, which bench.py quotes
# as roofline.traffic when workload and algorithmic bytes match).
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --cpu-steps 0 --streams 1 --roofline-steps 0 --no-extras"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/prof_traffic_$C.err
  echo "$C done"
done
timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-extras --roofline-steps 5 > $OUT/prof_traffic_bench.json 2> /dev/null
python3 - <<PY
import csv, glob, json, collections
def per_launch(counter):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % counter, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] == counter and n.startswith("void spmm_dw128_bf16_kernel"):
            agg["rna_1800x20000" if "<true" in n else "atac_900x100000"].append(float(r["Counter_Value"]))
    return {k: sum(v[-2:]) / len(v[-2:]) for k, v in agg.items()}
fe, wr = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
hbm = sum(2.0 * fe[k] * 1024 + wr[k] * 1024 for k in fe)
line = json.loads(open("$OUT/prof_traffic_bench.json").read().strip().splitlines()[-1])
out = {"kind": "spmm_dw_adamw",
       "kernel": "spmm_dw128_bf16_kernel<false / true, ADAM> (ATAC 900 x 100000, RNA hidden_rep + hidden_lib packed 1800 x 20000): first-layer weight gradient with the AdamW step in its epilogue",
       "workload": "C2 bench.py defaults (bf16, batch 1024): one launch per encoder and step",
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py $ARGS",
       "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming loads -> reads = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KiB",
       "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": line["roofline"]["algorithmic_bytes_per_step"],
       "ratio_to_algorithmic": hbm / line["roofline"]["algorithmic_bytes_per_step"]}
json.dump(out, open("$OUT/prof_c2_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
