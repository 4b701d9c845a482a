out=gpurun_out/r4c17
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $out/pytest_kernels.log
