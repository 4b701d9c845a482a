#!/bin/bash
out=gpurun_out/r3c14
mkdir -p $out
export TMPDIR=/tmp
L=mega/pytorch_amd/libmega_hip.so
for v in A B A B; do
  cp mega/pytorch_amd/libmega_hip_$v.so $L
  echo "== lib $v"; timeout 200 python tools/bench_kernels.py --frames 25 --what roi 2>&1 | grep roi_align
done
for v in A B; do
  cp mega/pytorch_amd/libmega_hip_$v.so $L
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b_$v.json 2> $out/b_$v.err
  echo "== lib $v"; grep "timed region" $out/b_$v.err; python - <<PY
import json
d=json.loads(open("$out/b_$v.json").read().strip().splitlines()[-1])
print(d["value"], [ (e["kernel"][:12], e["avg_launch_us"]) for e in d["roofline_hbm"] if "ROI" in e["kernel"] or "fc0" in e["kernel"]])
PY
done
cp mega/pytorch_amd/libmega_hip_B.so $L
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "roi" > $out/pytest_roi.log 2>&1; tail -3 $out/pytest_roi.log
cp mega/pytorch_amd/libmega_hip_E.so $L
timeout 600 python tools/gpu/timeline8.py --prebuilt > $out/timeline8.txt 2>&1; grep "==\|un-probed" $out/timeline8.txt
