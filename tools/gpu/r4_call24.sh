# round 4, call 24: frame stage as two 20-frame launches (layer3's conv3 working set fits the Infinity Cache at 20 frames:
# 4.6 against 3.5-3.8 TB/s), aggregation still 20 key frames
out=gpurun_out/r4c24
mkdir -p $out
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_FRAME_LAUNCH=20 timeout 300 python bench.py $b > $out/bench_f20.json 2> $out/bench_f20.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
MEGA_FRAME_LAUNCH=20 timeout 300 python bench.py $b > $out/bench_f20_2.json 2> $out/bench_f20_2.err
MEGA_FRAME_LAUNCH=10 timeout 300 python bench.py $b > $out/bench_f10.json 2> $out/bench_f10.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; grep -h "Error\|assert" $f | head -2; done
