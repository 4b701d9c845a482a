#!/bin/bash
out=gpurun_out/r3c2
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "r101_600x1000_f32" > $out/pytest_r101.log 2>&1; tail -25 $out/pytest_r101.log
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_e2e_gpu.py::test_r101_600x1000_f32_vs_oracle > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -15 $out/pytest_gpu.log
grep -E "ATTRIBUTION|^H f32|^F bf16|^B bf16|config [25]|R-101 600x1000|roi_align bf16|bf16 key frame|^bf16 " $out/pytest_gpu.log > $out/pytest_prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 20 > $out/bench_cli_spb20.json 2> $out/bench_cli_spb20.err; grep "timed region" $out/bench_cli_spb20.err
