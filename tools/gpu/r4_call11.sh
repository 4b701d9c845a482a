# round 4, call 11: step-batch structure inside a 20-step block now that the aggregation replays a hipGraph
out=gpurun_out/r4c11
mkdir -p $out
for spb in 20 10 5; do
  timeout 400 python bench.py --steps 20 --warmup 5 --steps-per-batch $spb --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_spb$spb.json 2> $out/bench_spb$spb.err
done
timeout 400 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > $out/bench_100.json 2> $out/bench_100.err
grep -h "timed region:" $out/*.err
