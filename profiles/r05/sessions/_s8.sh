#!/bin/bash
# round-5 session 8: whole GPU suite on the wave-per-bin emit kernel (+ its workgroup path at S = 40)
O=gpurun_out/r05_s8; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
