# round 4, call 25: layer3 over the two halves of the frame batch (Infinity-Cache-resident working set): bits + bench A/B
out=gpurun_out/r4c25
mkdir -p $out
MEGA_L3_SPLIT=1 timeout 600 python -m pytest tests/test_e2e_gpu.py -q -x -k "r101_600x1000_bf16 or batched_aggregation or engine_matches" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_L3_SPLIT=1 timeout 300 python bench.py $b > $out/bench_l3split.json 2> $out/bench_l3split.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
MEGA_L3_SPLIT=1 timeout 300 python bench.py $b > $out/bench_l3split2.json 2> $out/bench_l3split2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; grep -h "Error\|assert" $f | head -2; done
