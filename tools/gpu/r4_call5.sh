# round 4, call 5: attention micro-optimisations (tests + bench), steady-block timeline, per-kernel summary of a step-batch
out=gpurun_out/r4c5
mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or position or rounded_p or f32_stream" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -s -k "calibrated or attribution or graph_aggregation or long_clip_vs_reference_fixture or r101_600x1000_f32" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
grep -E "CALIBRATED|calibrated f32|ATTRIBUTION|^H " $out/pytest_e2e.log | cut -c1-420 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --head-stream bfloat16 > $out/bench_bf16stream.json 2> $out/bench_bf16stream.err
bash tools/gpu/trace_cli.sh r4c5/trace_cli > $out/trace_cli.log 2>&1
bash tools/gpu/trace.sh r4c5/trace > /dev/null 2>&1; python tools/trace_summary.py $out/trace/tail.csv > $out/trace_summary.txt 2>&1; head -3 $out/trace_summary.txt
grep -h "timed region:" $out/*.err
