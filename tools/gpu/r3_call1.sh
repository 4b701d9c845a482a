#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the new code, the driver-CLI bench (A/B: attention occupancy),
# and the per-layer kernel table.
out=gpurun_out/r3c1
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -s > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log
grep -E "ATTRIBUTION|^H f32|^F bf16|^B bf16|config [25]|R-101 600x1000|roi_align bf16|bf16 key frame" $out/pytest_gpu.log > $out/pytest_prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cli.json 2> $out/bench_cli.err; tail -4 $out/bench_cli.err; cut -c1-200 $out/bench_cli.json
MEGA_ATTN_OCC3=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_cli_occ3.json 2> $out/bench_cli_occ3.err; grep "timed region" $out/bench_cli_occ3.err
timeout 300 python tools/bench_kernels.py --frames 20 --what conv,attn,pos,roi > $out/kernels.txt 2>&1; tail -40 $out/kernels.txt
