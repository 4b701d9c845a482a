# round 4, call 22: one-segment / two-segment builds of the attention kernel with per-variant launch bounds
out=gpurun_out/r4c22
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "batched_aggregation or graph_aggregation or long_clip" > $out/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $out/pytest_e2e.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
bash tools/gpu/trace_cli.sh r4c22/trace_cli > /dev/null 2>&1; sed -n 1,3p gpurun_out/r4c22/trace_cli/cli_summary.txt
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r4c22/trace_cli/cli_tail.csv')))
K=[(int(r[0]),int(r[1]),r[3]) for r in rows]
pre=[i for i,r in enumerate(K) if 'stem_pool' in r[2]]
seg=K[pre[-1]:]
print([ (round((r[1]-r[0])/1e3,1), r[2][r[2].find('<'):r[2].find('>')+1]) for r in seg if 'attn_batched' in r[2]])
PY
