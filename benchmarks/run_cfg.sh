set -e
B="--cpu-steps 0 --steps 100 --no-extras --roofline-steps 0"
timeout -k 10 200 python3 bench.py $B --cov 4 > gpurun_out/r3_b5_cov.log 2> gpurun_out/r3_b5_cov.err
timeout -k 10 200 python3 bench.py $B --batchnorm > gpurun_out/r3_b5_bn.log 2> gpurun_out/r3_b5_bn.err
timeout -k 10 200 python3 bench.py $B --cov 4 --batchnorm --hidden 800 30 > gpurun_out/r3_b5_tut.log 2> gpurun_out/r3_b5_tut.err
BENCH_EXTRA="--cov 4 --batchnorm --hidden 800 30" bash benchmarks/kernel_times.sh > gpurun_out/r3_kt_tut.txt 2>&1
python3 - <<'PY'
import json
for t in ("cov","bn","tut"):
    d=json.loads(open(f"gpurun_out/r3_b5_{t}.log").read().strip().splitlines()[-1])
    print(t, round(d["value"]), d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
