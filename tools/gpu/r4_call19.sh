# round 4, call 19: first key segment at 32-aligned columns (pad rows in the projections): e2e bit identity, bench
out=gpurun_out/r4c19
mkdir -p $out
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "batched_aggregation or graph_aggregation or long_clip or early_position" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
MEGA_ATTN_SEGMENTS=0 timeout 300 python bench.py $b > $out/bench_copied_keysets.json 2> $out/bench_copied_keysets.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
bash tools/gpu/trace_cli.sh r4c19/trace_cli > /dev/null 2>&1; sed -n 1,12p gpurun_out/r4c19/trace_cli/cli_summary.txt
