#!/bin/bash
O=gpurun_out/r04_s8; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "many_bins or launch_shapes or full_size_properties" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s8 blockout general 2>&1 | grep irbpp | cut -c1-110
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_hfalways.so bash tools/gpu_kernel_stats.sh r04_s8/hfalways blockout 2>&1 | grep irbpp | cut -c1-110
