mkdir -p gpurun_out/c34
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "nms or rpn or postprocess" > gpurun_out/c34/pytest_k.log 2>&1; tail -3 gpurun_out/c34/pytest_k.log
timeout 300 python tools/gpu/nms_probe.py 2>&1 | grep "^pre" | cut -c1-60
