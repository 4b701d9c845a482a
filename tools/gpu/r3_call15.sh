#!/bin/bash
out=gpurun_out/r3c15
mkdir -p $out
export TMPDIR=/tmp
L=mega/pytorch_amd/libmega_hip.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or igemm8 or linear or first_fc or rpn_conv" > $out/pytest_k.log 2>&1; tail -3 $out/pytest_k.log
cp $L /tmp/prod.so
cp mega/pytorch_amd/libmega_hip_E.so $L
timeout 600 python tools/gpu/timeline8.py --prebuilt > $out/timeline8.txt 2>&1; grep "==\|un-probed\|slab\|tile total" $out/timeline8.txt | head -40
cp /tmp/prod.so $L
