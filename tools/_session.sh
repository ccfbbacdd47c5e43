O=gpurun_out/r2x; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
