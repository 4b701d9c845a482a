mkdir -p gpurun_out/c8; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -rf -k "roi or f32_end_to_end or bf16_end_to_end or C_dropins or r101 or long_clip or position" > gpurun_out/c8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c8/pytest.log
timeout 200 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c8/k_new.txt 2>&1
MEGA_ROI_NO_SEPARABLE=1 timeout 200 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c8/k_old.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c8/bA.json 2> gpurun_out/c8/bA.err
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 20 > gpurun_out/c8/bB.json 2> gpurun_out/c8/bB.err
tail -3 gpurun_out/c8/pytest.log; grep "timed region" gpurun_out/c8/b*.err; grep roi gpurun_out/c8/k_new.txt gpurun_out/c8/k_old.txt
