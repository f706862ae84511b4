set -e
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_trajectory_gpu.py tests/test_fullsize_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -1 gpurun_out/t_all.log
timeout -k 10 200 python bench.py --cpu-steps 0 2>&1 | grep '^{"metric"' | tail -1
