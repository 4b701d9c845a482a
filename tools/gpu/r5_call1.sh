#!/bin/bash
# round 5, call 1: the rewritten igemm8 epilogue (transposed accumulators, magic division) -- bit equality with the
# register-staged tiles + every conv test, the new SP kernels, then a same-box A/B of the bench against round 4's HEAD.
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -m gpu -k "conv or igemm8 or first_fc or linear or bottleneck or sp" > gpurun_out/r5c1/pytest_kernels.log 2>&1
tail -5 gpurun_out/r5c1/pytest_kernels.log
grep 'conv2d_sp x3\|linear_sp' gpurun_out/r5c1/pytest_kernels.log | head -20
bash tools/gpu/ab_bench.sh r5c1 2
