# round 4, call 9: downsample-variant fused bottleneck: bit equality, timing, bench A/B
out=gpurun_out/r4c9
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_bottleneck" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $out/pytest_kernels.log
timeout 300 python tools/gpu/bneck_bench.py > $out/bneck_bench.log 2>&1; cat $out/bneck_bench.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default.json 2> $out/bench_default.err
MEGA_FUSE_BOTTLENECK=id timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_id_only.json 2> $out/bench_id_only.err
MEGA_FUSE_BOTTLENECK=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_unfused.json 2> $out/bench_unfused.err
grep -h "timed region:" $out/*.err
