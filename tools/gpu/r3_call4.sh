#!/bin/bash
out=gpurun_out/r3c4
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv64 or multi_cat or conv2d_nhwc or relation_attention or position" > $out/pytest_k.log 2>&1; tail -5 $out/pytest_k.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "batched_aggregation or long_clip or reference_call or r101_600x1000_f32" > $out/pytest_e.log 2>&1; tail -5 $out/pytest_e.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
MEGA_CONV64=0 timeout 300 $B > $out/b_noconv64.json 2> $out/b_noconv64.err; grep "timed region" $out/b_noconv64.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
timeout 300 python tools/bench_kernels.py --frames 40 --what conv 2>&1 | head -8 > $out/kernels40.txt; cat $out/kernels40.txt
bash tools/gpu/trace.sh r3c4/trace > /dev/null 2>&1; python tools/trace_summary.py $out/trace/tail.csv > $out/trace_summary.txt 2>&1; head -32 $out/trace_summary.txt
