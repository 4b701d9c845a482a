mkdir -p gpurun_out/c9; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rf > gpurun_out/c9/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c9/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c9/bA.json 2> gpurun_out/c9/bA.err
timeout 300 python tools/bench_kernels.py --frames 20 --what conv,attn > gpurun_out/c9/kern.txt 2>&1
tail -3 gpurun_out/c9/pytest.log; grep "timed region" gpurun_out/c9/b*.err; grep "conv total\|attention core" gpurun_out/c9/kern.txt
