mkdir -p gpurun_out/c33
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "nms or rpn or postprocess" > gpurun_out/c33/pytest_k.log 2>&1; tail -3 gpurun_out/c33/pytest_k.log
bash tools/gpu/trace.sh c33
