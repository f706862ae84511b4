set -e
B="--cpu-steps 0 --steps 100 --no-extras --roofline-steps 0"
timeout -k 10 200 python3 bench.py $B > gpurun_out/r3_b6_plain.log 2> gpurun_out/r3_b6_plain.err
timeout -k 10 200 python3 bench.py $B --batchnorm > gpurun_out/r3_b6_bn.log 2> gpurun_out/r3_b6_bn.err
timeout -k 10 200 python3 bench.py $B --cov 4 --batchnorm --hidden 800 30 > gpurun_out/r3_b6_tut.log 2> gpurun_out/r3_b6_tut.err
python3 - <<'PY'
import json
for t in ("plain","bn","tut"):
    d=json.loads(open(f"gpurun_out/r3_b6_{t}.log").read().strip().splitlines()[-1])
    print(t, round(d["value"]), round(d["ms_per_step"],3), d["config"]["host_enqueue_ms_per_step"], d["config"]["final_losses"])
PY
