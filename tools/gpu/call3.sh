mkdir -p gpurun_out/c3; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -s -k "long_clip or batched or igemm8 or reference_call or static or conv" > gpurun_out/c3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c3/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c3/bA.json 2> gpurun_out/c3/bA.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --aggregation per-frame --no-roofline > gpurun_out/c3/bB.json 2> gpurun_out/c3/bB.err
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 20 > gpurun_out/c3/bC.json 2> gpurun_out/c3/bC.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-overlap > gpurun_out/c3/bD.json 2> gpurun_out/c3/bD.err
timeout 300 python tools/bench_kernels.py --frames 20 --what conv > gpurun_out/c3/conv20.txt 2>&1
tail -3 gpurun_out/c3/pytest.log; grep "timed region" gpurun_out/c3/b*.err; tail -1 gpurun_out/c3/conv20.txt
