# round 4, call 7: fused bottleneck v2 (transposed GEMMs) -- bit equality, timing
out=gpurun_out/r4c7
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_bottleneck" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $out/pytest_kernels.log
timeout 300 python tools/gpu/bneck_bench.py > $out/bneck_bench.log 2>&1; cat $out/bneck_bench.log
