#!/bin/bash
out=gpurun_out/r3c10
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or igemm8 or linear or first_fc or rpn_conv" > $out/pytest_k.log 2>&1; tail -4 $out/pytest_k.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "batched_aggregation or reference_call or record_reuse or static_aggregation_graph or r101_600x1000_f32 or long_clip" > $out/pytest_e.log 2>&1; tail -4 $out/pytest_e.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B > $out/b_default.json 2> $out/b_default.err; grep "timed region" $out/b_default.err
MEGA_IGEMM8_MIN_KTILES=2 timeout 300 $B --no-roofline > $out/b_minkt2.json 2> $out/b_minkt2.err; grep "timed region" $out/b_minkt2.err
timeout 300 $B --steps 100 --no-roofline > $out/b_100.json 2> $out/b_100.err; grep "timed region" $out/b_100.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c10/b_default.json"))
print(json.dumps(d["roofline"])[:600])
for e in d["roofline_hbm"]: print(e["kernel"][:60], e["achieved"], e["frac"], e["avg_launch_us"])
for k,v in list(d["kernel_families"].items())[:8]: print(k, v)
PY
