# round 4, call 2: calibrated fixture agreement, attribution with the split-precision FCs, conv64 bit test, bench A/B + f32 leg
out=gpurun_out/r4c2
mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -k "conv64 or f32_stream or split_v or cast" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -s -k "calibrated or attribution or batched_aggregation_is_bit or static_aggregation_graph" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
grep -E "ATTRIBUTION|^H |CALIBRATED|calibrated f32" $out/pytest_e2e.log | cut -c1-420 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --head-stream bfloat16 > $out/bench_bf16stream.json 2> $out/bench_bf16stream.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default2.json 2> $out/bench_default2.err
grep -h "timed region:\|f32 parity" $out/*.err
