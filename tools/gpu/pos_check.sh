#!/bin/bash
# position-logit kernel with per-box tables: parity tests, then same-box A/B of the bench against ab_old/
mkdir -p gpurun_out/pos
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "position or relation or attention" 2>&1 | tail -4 | tee gpurun_out/pos/tests.txt
bash tools/gpu/ab_bench.sh pos 2 2>&1 | tee gpurun_out/pos/ab.txt
