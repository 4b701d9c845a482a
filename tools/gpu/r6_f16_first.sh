# round 6, first GPU call of the fp16 mode: kernel tests, the seeded e2e leg, same-box bench bf16 vs f16
out=gpurun_out/r6_f16a
mkdir -p $out
timeout 1200 python -m pytest tests/test_f16_gpu.py -m gpu -q -x > $out/pytest_f16_kernels.log 2>&1; echo "f16 kernel tests rc=$?"; tail -15 $out/pytest_f16_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "f16_vs_oracle" > $out/pytest_f16_e2e.log 2>&1; echo "f16 e2e rc=$?"; grep -h "^f16 \|passed\|failed\|Error\|assert" $out/pytest_f16_e2e.log | head -20
for dt in bfloat16 float16 bfloat16 float16; do
  timeout 400 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-f32-leg --no-h2d-leg --no-whole-clip > $out/bench_$dt.json 2> $out/bench_$dt.err
  grep -h "timed region:" $out/bench_$dt.err | sed "s/^/$dt: /"
done
python - <<'PY'
import json
for dt in ("bfloat16", "float16"):
    try:
        l = json.loads(open("gpurun_out/r6_f16a/bench_%s.json" % dt).read())
        print(dt, l["value"], {k: v["ms_per_step"] for k, v in list(l["kernel_families"].items())[:12]})
    except Exception as e:
        print(dt, "no line", e)
PY
