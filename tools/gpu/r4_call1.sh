# round 4, call 1: the f32 head stream on the device -- kernel tests, attribution, bench A/B (head stream f32 vs bf16)
out=gpurun_out/r4c1
mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -k "cast or f32_stream or rounded_p or split_v or relation_attention or linear_split_k or first_fc" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -q -s -k "attribution or bf16_vs_oracle or batched_aggregation_is_bit or reference_call_convention or static_aggregation_graph or bf16_end_to_end" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
grep -E "ATTRIBUTION|^H |^F bf16|^B bf16|common-mode" $out/pytest_e2e.log $out/pytest_kernels.log | cut -c1-400 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --head-stream bfloat16 > $out/bench_bf16stream.json 2> $out/bench_bf16stream.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default2.json 2> $out/bench_default2.err
grep -h "timed region:\|f32 parity" $out/*.err
