# round 4, call 15: stem kernels without scratch and with launch bounds for 4 / 3 blocks per CU
out=gpurun_out/r4c15
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "stem" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $out/pytest_kernels.log
timeout 200 python tools/gpu/stem_bench.py 40 | tee $out/stem_bench.txt
timeout 200 python tools/gpu/stem_bench.py 20 | tee -a $out/stem_bench.txt
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_STEM_POOL=0 timeout 300 python bench.py $b > $out/bench_two_kernels.json 2> $out/bench_two_kernels.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
MEGA_STEM_POOL=0 timeout 300 python bench.py $b > $out/bench_two_kernels2.json 2> $out/bench_two_kernels2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
