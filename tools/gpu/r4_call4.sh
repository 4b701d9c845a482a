# round 4, call 4: u8 stem (bit test + bench), calibrated fixture, CLI block timeline
out=gpurun_out/r4c4
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "stem" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -s -k "calibrated or graph_aggregation or inference_loop_on_gpu or batched_aggregation_is_bit" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
grep -E "CALIBRATED|calibrated f32|graph aggregation:" $out/pytest_e2e.log | cut -c1-420 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --aggregation batched-eager > $out/bench_eager_agg.json 2> $out/bench_eager_agg.err
bash tools/gpu/trace_cli.sh r4c4/trace_cli > $out/trace_cli.log 2>&1
TRACE_ARGS="--aggregation batched-eager" bash tools/gpu/trace_cli.sh r4c4/trace_cli_eager > $out/trace_cli_eager.log 2>&1
grep -h "timed region:" $out/*.err
