# round 4, call 13: does a high-priority aggregation stream make the in-block overlap of [10, 10] reliable?
out=gpurun_out/r4c13
mkdir -p $out
b="--warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py --steps 20 $b > $out/bench_1010.json 2> $out/bench_1010.err
MEGA_AGG_PRIORITY=1 timeout 300 python bench.py --steps 20 $b > $out/bench_1010_prio.json 2> $out/bench_1010_prio.err
timeout 300 python bench.py --steps 20 --steps-per-batch 20 $b > $out/bench_20.json 2> $out/bench_20.err
MEGA_AGG_PRIORITY=1 timeout 300 python bench.py --steps 20 --steps-per-batch 20 $b > $out/bench_20_prio.json 2> $out/bench_20_prio.err
timeout 300 python bench.py --steps 100 $b > $out/bench_100.json 2> $out/bench_100.err
MEGA_AGG_PRIORITY=1 timeout 300 python bench.py --steps 100 $b > $out/bench_100_prio.json 2> $out/bench_100_prio.err
timeout 300 python bench.py --steps 20 $b > $out/bench_1010_again.json 2> $out/bench_1010_again.err
timeout 300 python bench.py --steps 20 --steps-per-batch 20 $b > $out/bench_20_again.json 2> $out/bench_20_again.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
