# round 4, call 23: copy_any_kernel (all copy segments of a call in one launch at any alignment): tests, bench A/B
out=gpurun_out/r4c23
mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "cat or copy or assemble or cast" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "batched_aggregation or graph_aggregation or long_clip or shard" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_COPY_ANY=0 timeout 300 python bench.py $b > $out/bench_per_width_copies.json 2> $out/bench_per_width_copies.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
timeout 300 python bench.py $b --aggregation batched-eager > $out/bench_eager.json 2> $out/bench_eager.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
bash tools/gpu/trace_cli.sh r4c23/trace_cli > /dev/null 2>&1; sed -n 1,3p gpurun_out/r4c23/trace_cli/cli_summary.txt; grep "copy_\|CatArray" gpurun_out/r4c23/trace_cli/cli_summary.txt
