#!/bin/bash
source benchmarks/gpu_steps.sh
step r3_t_epoch 900 python -u -m pytest tests/test_epoch_control_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q -s --timeout 600 --timeout-method thread -k "epoch_control or whole_paired or tail_rows or wgrad"
step r3_b4_cov 300 python bench.py --steps 100 --cpu-steps 0 --no-extras --cov 4
step r3_b4_bn 300 python bench.py --steps 100 --cpu-steps 0 --no-extras --batchnorm
step r3_b4_tut 300 python bench.py --steps 100 --cpu-steps 0 --no-extras --batchnorm --cov 4 --hidden 800 30
