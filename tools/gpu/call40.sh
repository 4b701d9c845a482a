mkdir -p gpurun_out/c40
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q > gpurun_out/c40/pytest_k.log 2>&1; tail -3 gpurun_out/c40/pytest_k.log
timeout 300 python tools/gpu/stem_probe.py 2>&1 | tail -5
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c40/b20.json 2> gpurun_out/c40/b20.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c40/b100.json 2> gpurun_out/c40/b100.err
grep -h "timed region:" gpurun_out/c40/*.err
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/c40/pytest.log 2>&1; tail -3 gpurun_out/c40/pytest.log
