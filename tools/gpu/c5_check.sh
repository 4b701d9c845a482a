#!/bin/bash
# config 5 (FGFA) work of round 6: the new kernels' tests, the FGFA end-to-end tests, the per-layer table and the bench line
mkdir -p gpurun_out/c5
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "deconv_subpixel or caller_split_k" 2>&1 | tail -5 | tee gpurun_out/c5/tests_new.txt
python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "fgfa or flownet or cfg5 or dff" -s 2>&1 | tail -30 | tee gpurun_out/c5/tests_fgfa.txt
python tools/gpu/flownet_layers.py > gpurun_out/c5/layers.txt 2>&1
python tools/bench_configs.py --config 5 --no-cpu-baseline > gpurun_out/c5/line.json 2> gpurun_out/c5/err.txt
cat gpurun_out/c5/line.json | head -c 600
