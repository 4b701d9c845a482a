MEGA_IGEMM8_MIN_KTILES=1 timeout 300 python tools/gpu/igemm8_check.py --quick 2>&1 | tail -9
MEGA_IGEMM8_MIN_KTILES=1 timeout 300 python tools/bench_kernels.py --frames 20 --what conv --tiles 8:256,8:192,128x128 2>&1 | grep "K=    64\|K=   128"
