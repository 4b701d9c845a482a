#!/bin/bash
out=gpurun_out/r3c9
mkdir -p $out
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "fgfa or postprocess" > $out/pytest.log 2>&1; tail -8 $out/pytest.log
