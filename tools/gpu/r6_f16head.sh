# round 6: the fp16 head -- calibrated fixture legs + the bench line with the new bf16x3 + fp16-head leg
out=gpurun_out/${1:-r6_f16head}
mkdir -p $out
timeout 1200 python -m pytest tests/test_e2e_gpu.py -q -x -s -k "calibrated_bf16_agreement" > $out/calibrated.log 2>&1; echo "pytest rc=$?"; grep "CALIBRATED X3\|passed\|failed\|Error" $out/calibrated.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
grep -h "timed region:\|leg\|head" $out/bench.err | cut -c1-220
