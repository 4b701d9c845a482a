#!/bin/bash
out=gpurun_out/r3c12
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $out/pytest_k.log 2>&1; tail -3 $out/pytest_k.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
timeout 300 python tools/bench_kernels.py --frames 40 --what conv > $out/kernels40.txt 2>&1; grep "l3.conv3\|l3.conv1\|r5.conv3\|l1.conv3\|l2.conv3\|total" $out/kernels40.txt
grep -E "l1\.|l2\.|stem|pool" $out/kernels40.txt | head -30
