#!/bin/bash
# round 5, call 11: BASELINE configs[1] (MEGA R-50 fp32) in the split-precision mode: parity pin + bench line next to exact f32
export TMPDIR=/tmp
out=gpurun_out/r5c11
mkdir -p $out
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "cfg2" > $out/e.log 2>&1; tail -3 $out/e.log; grep -E '^config 2' $out/e.log
timeout 300 python tools/bench_configs.py --config 2 > $out/config2.json 2> $out/config2.err; cut -c1-200 $out/config2.json
timeout 300 python tools/bench_configs.py --config 2 --f32-conv bf16x3 > $out/config2_bf16x3.json 2> $out/config2_bf16x3.err; cut -c1-260 $out/config2_bf16x3.json; tail -3 $out/config2_bf16x3.err
