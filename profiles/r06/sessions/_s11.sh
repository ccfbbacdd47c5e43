#!/bin/bash
# round-6 session 11: placement-steps/s of the wide capacity path (32 x 32 action grid, resolutionA = 0.01), scripted MINZ policy fused
O=gpurun_out/r06_s11; mkdir -p $O
timeout 900 python tools/wide_throughput.py 2>&1 | tail -1 | tee $O/wide_throughput.json
