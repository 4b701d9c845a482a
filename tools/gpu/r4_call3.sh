# round 4, call 3: graph aggregation (bit identity + bench A/B), calibrated fixture with scale-aware f32 tolerance
out=gpurun_out/r4c3
mkdir -p $out
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -s -k "graph_aggregation or calibrated or long_clip_vs_reference_fixture" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
grep -E "CALIBRATED|calibrated f32|graph aggregation:" $out/pytest_e2e.log | cut -c1-420 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --aggregation batched-eager > $out/bench_eager_agg.json 2> $out/bench_eager_agg.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --head-stream bfloat16 > $out/bench_bf16stream.json 2> $out/bench_bf16stream.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_100.json 2> $out/bench_100.err
grep -h "timed region:\|f32 parity" $out/*.err
