#!/bin/bash
# the split-precision mode's ROIAlign in the separable form: kernel test, the mode's parity tests, same-box A/B of its bench leg
mkdir -p gpurun_out/x3roi
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "roi_align" 2>&1 | tail -3
python -m pytest tests/test_e2e_gpu.py tests/test_f16_gpu.py -x -q -m gpu -s -k "bf16x3 or x3 or calibrated or f16_head" 2>&1 | grep -E "passed|failed|X3|bf16x3 |X3H" | cut -c1-260 | tail -24
leg="--steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --min-seconds 2"
for i in 1 2; do
  MEGA_ROI_NO_SEPARABLE=1 python bench.py $leg 2>&1 >/dev/null | grep -o "timed region.*frames/s)" | sed 's/^/exact-order ROIAlign: /'
  python bench.py $leg 2>&1 >/dev/null | grep -o "timed region.*frames/s)" | sed 's/^/separable ROIAlign:   /'
done | tee gpurun_out/x3roi/ab.txt
