#!/bin/bash
out=gpurun_out/r3c17
mkdir -p $out
export TMPDIR=/tmp
L=mega/pytorch_amd/libmega_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $out/pytest_k.log 2>&1; tail -3 $out/pytest_k.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
timeout 300 python tools/bench_kernels.py --frames 40 --what conv > $out/kernels40.txt 2>&1; grep "l3.conv\|r5.conv3\|l1.conv3\|l2.conv3\|rpn.conv\|total" $out/kernels40.txt
cp $L /tmp/prod.so
cp mega/pytorch_amd/libmega_hip_E.so $L
timeout 600 python tools/gpu/timeline8.py --prebuilt > $out/timeline8.txt 2>&1; grep "==\|un-probed\|slab\|tile total" $out/timeline8.txt | head -24
cp /tmp/prod.so $L
