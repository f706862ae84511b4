# The paired step in the fp32-equivalent precision mode (the package default; every MFMA operand split into 3 bf16 terms):
# bench line + per-kernel statistics of a single-stream run
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
timeout -k 10 300 python3 bench.py --precision fp32 --cpu-steps 0 --steps 100 --warmup 10 --no-extras > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/fp32prof -o fp32 -- python3 bench.py --precision fp32 --streams 1 --cpu-steps 0 --steps 30 --warmup 5 --roofline-steps 0 --no-extras > /dev/null 2>&1
f=$(find gpurun_out/fp32prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/fp32_kernel_stats.csv
rm -rf gpurun_out/fp32prof
