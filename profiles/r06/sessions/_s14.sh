#!/bin/bash
# round-6 session 14: phase stamps of the wide kernel
O=gpurun_out/r06_s14; mkdir -p $O
timeout 600 python tools/wide_phase_profile.py 2>&1 | tail -1 | tee $O/wide_phases.json
