# Rehearsal of the N > 1 bench paths on a one-GPU box: two ranks over gloo through host memory (timings are not xGMI
# figures), then ONE rank over RCCL with the feature-sharded call sequence (reduce-scatter, all-gather, all-reduce)
set -e
mkdir -p gpurun_out
B="--steps 8 --warmup 3 --cpu-steps 0 --roofline-steps 4 --cells 20000"
SCPAIR_BENCH_WATCHDOG=250 timeout -k 10 300 python3 bench.py --gpus 2 $B --replicated > gpurun_out/r3_dp2_repl.log 2> gpurun_out/r3_dp2_repl.err
SCPAIR_BENCH_WATCHDOG=250 timeout -k 10 300 python3 bench.py --gpus 2 $B > gpurun_out/r3_dp2_shard.log 2> gpurun_out/r3_dp2_shard.err
SCPAIR_BENCH_FORCE_DIST=1 SCPAIR_BENCH_WATCHDOG=250 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-steps 0 --shard-first-layer > gpurun_out/r3_rccl1_shard.log 2> gpurun_out/r3_rccl1_shard.err
SCPAIR_BENCH_FORCE_DIST=1 SCPAIR_BENCH_WATCHDOG=250 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-steps 0 --replicated > gpurun_out/r3_rccl1_repl.log 2> gpurun_out/r3_rccl1_repl.err
python3 - <<'PY'
import json
for t in ("dp2_repl", "dp2_shard", "rccl1_shard", "rccl1_repl"):
    try:
        d = json.loads(open(f"gpurun_out/r3_{t}.log").read().strip().splitlines()[-1])
        print(t, round(d["value"]), round(d["ms_per_step"], 3), d["n_gpus"], d["config"]["first_layer"], d["config"]["final_losses"], round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(t, "ERR", e)
PY
