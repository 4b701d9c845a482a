mkdir -p gpurun_out/c12; export TMPDIR=/tmp
timeout 600 python tools/gpu/igemm8_check.py > gpurun_out/c12/check.txt 2>&1; echo "check rc=$?" >> gpurun_out/c12/check.txt
timeout 600 python tools/bench_kernels.py --frames 20 --what conv > gpurun_out/c12/conv20.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c12/bA.json 2> gpurun_out/c12/bA.err
tail -2 gpurun_out/c12/check.txt; grep "timed region" gpurun_out/c12/b*.err; grep "l3.conv\|rpn.conv\|r5\|fc0\|conv total" gpurun_out/c12/conv20.txt
