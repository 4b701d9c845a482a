mkdir -p gpurun_out/c6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -rf -k "roi or position or relation or long_clip or batched or C_dropins or r101" > gpurun_out/c6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c6/pytest.log
timeout 200 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c6/k_new.txt 2>&1
MEGA_POS_LEGACY=1 MEGA_ROI_NO_XCD_SLICE=1 timeout 200 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c6/k_old.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c6/bA.json 2> gpurun_out/c6/bA.err
MEGA_POS_LEGACY=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c6/bB.json 2> gpurun_out/c6/bB.err
tail -3 gpurun_out/c6/pytest.log; grep "timed region" gpurun_out/c6/b*.err; tail -2 gpurun_out/c6/k_new.txt gpurun_out/c6/k_old.txt
