# round 4, call 12: position logits of all stages ahead of time on a side stream: bit identity (batched vs per-frame, graph vs
# eager, sharded), bench A/B
out=gpurun_out/r4c12
mkdir -p $out
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "batched_aggregation or graph_aggregation or shard or static" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_default.json 2> $out/bench_default.err
MEGA_EARLY_POS=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_inline_pos.json 2> $out/bench_inline_pos.err
timeout 400 python bench.py --steps 20 --warmup 5 --steps-per-batch 20 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_spb20.json 2> $out/bench_spb20.err
MEGA_EARLY_POS=0 timeout 400 python bench.py --steps 20 --warmup 5 --steps-per-batch 20 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_spb20_inline_pos.json 2> $out/bench_spb20_inline_pos.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_default2.json 2> $out/bench_default2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c1-150)"; grep -h "Error\|error" $f | head -3; done
TRACE_ARGS="" bash tools/gpu/trace_cli.sh r4c12/trace_cli > /dev/null 2>&1; head -8 gpurun_out/r4c12/trace_cli/cli_summary.txt
