#!/bin/bash
out=gpurun_out/r3c19
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
bash tools/gpu/trace_cli.sh r3c19 | head -12
