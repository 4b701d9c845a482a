mkdir -p gpurun_out/c14
timeout 600 python tools/gpu/igemm8_check.py --quick > gpurun_out/c14/check.txt 2>&1; echo "check rc=$?" >> gpurun_out/c14/check.txt
timeout 300 python tools/gpu/ablate8.py > gpurun_out/c14/ablate.txt 2>&1
timeout 600 python tools/bench_kernels.py --frames 20 --what conv > gpurun_out/c14/conv20.txt 2>&1
tail -2 gpurun_out/c14/check.txt; grep -v amdgpu gpurun_out/c14/ablate.txt; grep "l3.conv\|rpn.conv\|r5\|fc0\|conv total" gpurun_out/c14/conv20.txt
