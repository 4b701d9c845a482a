mkdir -p gpurun_out/c21
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q -rf -k "first_fc or split_k or igemm8 or r101 or long_clip" > gpurun_out/c21/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c21/pytest.log
timeout 300 python tools/bench_kernels.py --frames 20 --what conv --tiles 8:256,8:192,128x128 2>&1 | grep "fc0\|conv total" > gpurun_out/c21/fc0.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c21/b20.json 2> gpurun_out/c21/b20.err
tail -3 gpurun_out/c21/pytest.log; cat gpurun_out/c21/fc0.txt; grep "timed region" gpurun_out/c21/*.err
