#!/bin/bash
source benchmarks/gpu_steps.sh
step r3_t_epoch 900 python -u -m pytest tests/test_epoch_control_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -s --timeout 600 --timeout-method thread -k "epoch_control or whole_paired or tail_rows or wgrad"
