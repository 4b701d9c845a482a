mkdir -p gpurun_out/c7; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf > gpurun_out/c7/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c7/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bA.json 2> gpurun_out/c7/bA.err
for c in 1 2 5; do timeout 300 python tools/bench_configs.py --config $c > gpurun_out/c7/config$c.json 2> gpurun_out/c7/config$c.err; done
timeout 200 python tools/bench_configs.py --config 1 --dtype float32 > gpurun_out/c7/config1_f32.json 2> gpurun_out/c7/config1_f32.err
tail -3 gpurun_out/c7/pytest.log; grep "timed region" gpurun_out/c7/b*.err; cut -c1-300 gpurun_out/c7/config*.json; tail -2 gpurun_out/c7/config*.err
