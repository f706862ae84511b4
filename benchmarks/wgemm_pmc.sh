#!/bin/bash
# SQ counters of the wide GEMM kernel (one shape, one round of 256 workgroups), with and without operand traffic.
# usage on the GPU box: bash benchmarks/wgemm_pmc.sh   -> gpurun_out/wgemm_pmc.txt
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
export SCPAIR_WGEMM_NSPLIT=1
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
rm -f $ROOT/gpurun_out/wgemm_pmc.txt
for dbg in 0 1; do
  export SCPAIR_WGEMM_DBG=$dbg
  timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d /tmp/wpmc$dbg -o p -- python3 benchmarks/wgemm_probe.py 0 0 1024 16384 20000 > /dev/null 2>&1
  python3 - <<PY >> $ROOT/gpurun_out/wgemm_pmc.txt
import csv, glob, collections
f = glob.glob("/tmp/wpmc$dbg/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "wgemm_kernel" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
print("SCPAIR_WGEMM_DBG=$dbg (1: no operand traffic)  1024 x 16384 x 20000, one K-range, eight waves")
for k in "$CNT".split():
    print(f"  {k:28s} {m.get(k, 0):14.4g}")
print(f"  MFMA busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.3f}   LDS bank conflict cycles / LDS active = {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_ACTIVE_INST_LDS'], 1):.3f}")
PY
done
cat $ROOT/gpurun_out/wgemm_pmc.txt
