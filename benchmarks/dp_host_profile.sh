# Host-side profile of the N > 1 step on a one-GPU box: ONE rank over RCCL (the call sequence of the feature-sharded and the
# replicated scheme), cProfile of the timed steps written next to the bench line
set -e
mkdir -p gpurun_out
for mode in shard repl; do
  if [ $mode = shard ]; then flag=--shard-first-layer; else flag=--replicated; fi
  SCPAIR_BENCH_CPROFILE=gpurun_out/dp_host_$mode.txt SCPAIR_BENCH_CPROFILE_SORT=tottime SCPAIR_BENCH_FORCE_DIST=1 SCPAIR_BENCH_WATCHDOG=250 \
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --steps 200 --warmup 20 --cpu-steps 0 --no-extras $flag > gpurun_out/dp_host_$mode.log 2> gpurun_out/dp_host_$mode.err
  tail -1 gpurun_out/dp_host_$mode.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', round(d['value']), round(d['ms_per_step'],3))"
done
