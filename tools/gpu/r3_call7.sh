#!/bin/bash
out=gpurun_out/r3c7
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q > $out/pytest_e2e.log 2>&1; tail -6 $out/pytest_e2e.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region\|host ms" $out/b_default.err
timeout 300 $B --steps-per-batch 10 > $out/b_spb10.json 2> $out/b_spb10.err; grep "timed region" $out/b_spb10.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
MEGA_FORCE_SHARDED=1 timeout 300 $B > $out/b_sharded.json 2> $out/b_sharded.err; grep "timed region" $out/b_sharded.err
timeout 400 python tools/bench_configs.py --config 5 > $out/config5.json 2> $out/config5.err; tail -2 $out/config5.err; cut -c1-120 $out/config5.json
