#!/bin/bash
out=gpurun_out/r3c6
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "fgfa" > $out/pytest_fgfa.log 2>&1; tail -15 $out/pytest_fgfa.log
timeout 400 python tools/bench_configs.py --config 5 > $out/config5.json 2> $out/config5.err; tail -3 $out/config5.err; cut -c1-700 $out/config5.json
