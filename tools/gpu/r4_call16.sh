# round 4, call 16: attention with two key segments read in place: kernel bit equality, e2e bit identity, bench A/B
out=gpurun_out/r4c16
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or attn" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "batched_aggregation or graph_aggregation or early_position or attribution or long_clip" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_ATTN_SEGMENTS=0 timeout 300 python bench.py $b > $out/bench_copied_keysets.json 2> $out/bench_copied_keysets.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
MEGA_ATTN_SEGMENTS=0 timeout 300 python bench.py $b > $out/bench_copied_keysets2.json 2> $out/bench_copied_keysets2.err
MEGA_ATTN_OCC3=0 timeout 300 python bench.py $b > $out/bench_occ2.json 2> $out/bench_occ2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; grep -h "Error" $f | head -2; done
