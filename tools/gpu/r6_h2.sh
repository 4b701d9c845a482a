out=gpurun_out/r6_h2
mkdir -p $out
timeout 900 python -m pytest tests/test_f16_gpu.py -m gpu -q -x -k "h2" > $out/pytest_h2_kernels.log 2>&1; echo "h2 kernel tests rc=$?"; tail -12 $out/pytest_h2_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "f16x2_vs_oracle or f16_vs_oracle" > $out/pytest_h2_e2e.log 2>&1; echo "e2e rc=$?"; grep -h "^f16\|passed\|failed\|Error\|assert" $out/pytest_h2_e2e.log | head -20
timeout 900 python -m pytest tests/test_feed.py -m gpu -q -x -k "read_ahead" 2>&1 | tail -5
for dt in f16x2 float16; do
  timeout 400 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-f32-leg --no-whole-clip > $out/bench_$dt.json 2> $out/bench_$dt.err
  grep -h "timed region:\|with H2D" $out/bench_$dt.err | sed "s/^/$dt: /"
done
python - <<'PY'
import json
for dt in ("f16x2", "float16"):
    try:
        l = json.loads(open("gpurun_out/r6_h2/bench_%s.json" % dt).read())
        print(dt, l["value"], {k: v["ms_per_step"] for k, v in list(l["kernel_families"].items())[:10]})
        print("   with_h2d:", {k: l["config"]["with_h2d"][k] for k in ("fps", "vs_resident")} if l["config"]["with_h2d"] else None)
    except Exception as e:
        print(dt, "no line", e)
PY
