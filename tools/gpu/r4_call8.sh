# round 4, call 8: config pins (cfg1 / cfg5 f32 + bf16), fused cat+cast, forced-sharded bench on one GPU, default bench
out=gpurun_out/r4c8
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "cat_rows_cast or multi_cat" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -s -k "cfg1 or cfg5 or batched_aggregation_is_bit or graph_aggregation" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $out/pytest_e2e.log
grep -E "config 1|config 5|cfg1 base|cfg5 FGFA" $out/pytest_e2e.log | cut -c1-420 > $out/prints.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default.json 2> $out/bench_default.err
MEGA_FORCE_SHARDED=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_forced_sharded.json 2> $out/bench_forced_sharded.err
grep -h "timed region:" $out/*.err
tail -5 $out/bench_forced_sharded.err | cut -c1-300
