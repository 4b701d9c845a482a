timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "stem" 2>&1 | tail -3
timeout 300 python tools/gpu/stem_probe.py 2>&1 | tail -5
