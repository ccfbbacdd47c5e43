#!/bin/bash
# round-5 session 36: tests of the grouped env's policy over all bins, make_vec_envs' one-group default, per-group D2H on the groups' own streams
O=gpurun_out/r05_s36; mkdir -p $O
timeout 600 python -m pytest tests/test_grouped.py tests/test_gpu_boundary.py tests/test_gpu_features.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
