#!/bin/bash
# round 5, call 4: ROIAlign writing planes (bit-equality), the bf16x3 mode's seeded-fixture metrics, its speed
export TMPDIR=/tmp
out=gpurun_out/r5c4
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "roi_align_planes or sp" > $out/pytest_kernels.log 2>&1
tail -2 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "bf16x3_vs_oracle" > $out/pytest_e2e.log 2>&1
tail -3 $out/pytest_e2e.log
grep -E '^bf16x3 ' $out/pytest_e2e.log
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_x3.json 2> $out/bench_x3.err
grep -E 'timed region|skipped|rror' $out/bench_x3.err | head -5
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c4/bench_x3.json"))
print("x3:", d["value"], "fps")
for k, v in d["kernel_families"].items():
    print("  %-32s %s" % (k, v))
PY
