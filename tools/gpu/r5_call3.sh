#!/bin/bash
# round 5, call 3: the bf16x3 mode's kernel-family breakdown (narrow layers skip out-of-range MFMAs now), its e2e test with
# the stated permutation allowance, the forced-sharded bit-equality test, the with_h2d leg with the chunked staging pipeline.
export TMPDIR=/tmp
out=gpurun_out/r5c3
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sp" > $out/pytest_kernels.log 2>&1
tail -2 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "bf16x3_vs_oracle or forced_sharded" > $out/pytest_e2e.log 2>&1
tail -5 $out/pytest_e2e.log
grep -E 'bf16x3 key frame' $out/pytest_e2e.log
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_x3.json 2> $out/bench_x3.err
grep -E 'timed region|skipped|rror' $out/bench_x3.err | head -5
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c3/bench_x3.json"))
print("x3:", d["value"], "fps")
for k, v in d["kernel_families"].items():
    print("  %-32s %s" % (k, v))
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-seconds 3 > $out/bench.json 2> $out/bench.err
grep -E 'timed region|H2D|skipped' $out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r5c3/bench.json')); print(json.dumps(d['config']['with_h2d'])[:600])"
