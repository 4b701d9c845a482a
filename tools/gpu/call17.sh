mkdir -p gpurun_out/c17
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -rf -k "roi or C_dropins or r101 or long_clip" > gpurun_out/c17/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c17/pytest.log
timeout 100 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c17/roi.txt 2>&1
MEGA_ROI_NO_XCD_SLICE=1 timeout 100 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c17/roi_noslice.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c17/bA.json 2> gpurun_out/c17/bA.err
tail -3 gpurun_out/c17/pytest.log; grep roi_align gpurun_out/c17/roi*.txt; grep "timed region" gpurun_out/c17/bA.err
