mkdir -p gpurun_out/c10; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -rf -k "roi or relation or attention or long_clip or r101 or C_dropins or batched or full_size" > gpurun_out/c10/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c10/pytest.log
timeout 300 python tools/bench_kernels.py --frames 20 --what attn,roi > gpurun_out/c10/kern.txt 2>&1
MEGA_ROI_NO_SEPARABLE=1 timeout 100 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c10/roi_old.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c10/bA.json 2> gpurun_out/c10/bA.err
tail -3 gpurun_out/c10/pytest.log; grep "timed region" gpurun_out/c10/b*.err; grep "roi_align\|attention core" gpurun_out/c10/kern.txt gpurun_out/c10/roi_old.txt
