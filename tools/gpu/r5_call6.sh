#!/bin/bash
# round 5, call 6: the head's linears in split precision (conv_mode x3): kernel test, both fixtures, speed with / without
export TMPDIR=/tmp
out=gpurun_out/r5c6
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "x3_weight or sp" > $out/pytest_kernels.log 2>&1
tail -2 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "bf16x3_vs_oracle or calibrated" > $out/pytest_e2e.log 2>&1
tail -3 $out/pytest_e2e.log
grep -E '^bf16x3 |CALIBRATED X3' $out/pytest_e2e.log
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-f32-leg --no-h2d-leg --no-roofline --min-seconds 2 > $out/bench_x3.json 2> $out/bench_x3.err
grep -E 'timed region|skipped|rror' $out/bench_x3.err | head -5
