mkdir -p gpurun_out/c2; export TMPDIR=/tmp
timeout 600 python tools/gpu/igemm8_check.py > gpurun_out/c2/check.txt 2>&1; echo "check rc=$?" >> gpurun_out/c2/check.txt
timeout 600 python tools/bench_kernels.py --frames 20 --what conv --tiles 128x128,8:256,8:192 > gpurun_out/c2/conv20.txt 2>&1
timeout 300 python tools/bench_kernels.py --frames 26 --what conv --tiles 128x128,8:256 > gpurun_out/c2/conv26.txt 2>&1
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -s -k "long_clip or r101 or conv or hot or linear" > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c2/bA.json 2> gpurun_out/c2/bA.err
tail -3 gpurun_out/c2/check.txt; tail -3 gpurun_out/c2/pytest.log; grep "timed region" gpurun_out/c2/b*.err; tail -2 gpurun_out/c2/conv20.txt
