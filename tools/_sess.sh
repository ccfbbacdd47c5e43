#!/bin/bash
bash tools/gpu_session.sh r03w test smoke workloads
