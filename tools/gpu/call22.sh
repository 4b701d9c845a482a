mkdir -p gpurun_out/c22
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -k "long_clip or batched or reference_call or r101 or relation or position" > gpurun_out/c22/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c22/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c22/b20.json 2> gpurun_out/c22/b20.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c22/b100.json 2> gpurun_out/c22/b100.err
tail -4 gpurun_out/c22/pytest.log; grep "timed region" gpurun_out/c22/*.err
