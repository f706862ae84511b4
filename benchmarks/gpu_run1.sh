set -e
SCPAIR_BENCH_CPROFILE_SORT=tottime SCPAIR_BENCH_CPROFILE=gpurun_out/cprof_tot.txt timeout -k 10 200 python bench.py --gpus 1 --steps 30 --warmup 5 --cpu-steps 0 > gpurun_out/b_single.log 2>&1
tail -1 gpurun_out/b_single.log | cut -c1-200
