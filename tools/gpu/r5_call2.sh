#!/bin/bash
# round 5, call 2: SP kernels (f32-out + residual fixed), the bf16x3 / wide modes end to end against the oracle fixtures,
# the bench with its new legs (bf16x3 parity mode, with_h2d) and the wide-trunk mode timed.
export TMPDIR=/tmp
out=gpurun_out/r5c2
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -m gpu -k "sp or stem" > $out/pytest_kernels.log 2>&1
tail -3 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "bf16x3_vs_oracle or r101_bf16_attribution or calibrated" > $out/pytest_e2e.log 2>&1
tail -5 $out/pytest_e2e.log
grep -E 'bf16x3 key frame|^W wide|^X bf16x3|^B bf16|^F bf16|CALIBRATED (W|X3|B)' $out/pytest_e2e.log | head -60
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
grep -E 'timed region|leg|H2D|skipped' $out/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --dtype wide --no-cpu-baseline --no-f32-leg --no-h2d-leg --no-roofline --min-seconds 3 > $out/bench_wide.json 2> $out/bench_wide.err
grep -E 'timed region|skipped|Error|error' $out/bench_wide.err | head
