set -e
SCPAIR_BENCH_FORCE_DIST=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-steps 0 > gpurun_out/b_rccl1.log 2>&1 || { tail -20 gpurun_out/b_rccl1.log; exit 1; }
grep '^{"metric"' gpurun_out/b_rccl1.log | tail -1
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-fuse-opt > gpurun_out/b_nf.log 2>&1
grep '^{"metric"' gpurun_out/b_nf.log | tail -1
