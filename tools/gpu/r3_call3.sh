#!/bin/bash
out=gpurun_out/r3c3
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -k "position or attention or batched_aggregation or long_clip" > $out/pytest_rel.log 2>&1; tail -3 $out/pytest_rel.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
MEGA_ATTN_BLOCKS=64 timeout 300 $B > $out/b_blocks64.json 2> $out/b_blocks64.err; grep "timed region" $out/b_blocks64.err
MEGA_ATTN_OCC3=1 timeout 300 $B > $out/b_occ3.json 2> $out/b_occ3.err; grep "timed region" $out/b_occ3.err
MEGA_ATTN_BLOCKS=64 MEGA_ATTN_OCC3=1 timeout 300 $B > $out/b_blocks64_occ3.json 2> $out/b_blocks64_occ3.err; grep "timed region" $out/b_blocks64_occ3.err
timeout 300 $B --steps 100 > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
bash tools/gpu/trace.sh r3c3/trace > /dev/null 2>&1; python tools/trace_summary.py $out/trace/tail.csv > $out/trace_summary.txt 2>&1; sed -n 18,30p $out/trace_summary.txt
