mkdir -p gpurun_out/c41
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "roi" > gpurun_out/c41/pytest_k.log 2>&1; tail -2 gpurun_out/c41/pytest_k.log
timeout 300 python tools/bench_kernels.py --frames 20 --what roi 2>&1 | tail -4
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c41/b20.json 2> gpurun_out/c41/b20.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c41/b100.json 2> gpurun_out/c41/b100.err
grep -h "timed region:" gpurun_out/c41/*.err
