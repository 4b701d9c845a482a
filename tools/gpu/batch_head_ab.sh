#!/bin/bash
python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "dff_engine or fgfa_engine_equals or base_engine" 2>&1 | tail -2
mkdir -p gpurun_out/c5
for bh in 0 1; do
  python tools/bench_configs.py --config 1 --no-cpu-baseline --batch-head $bh 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base  batch_head=$bh: engine %.1f FPS (call convention %.1f)' % (d['config']['clip_engine']['fps'], d['value']))"
  python tools/bench_configs.py --method dff --batch-head $bh 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dff   batch_head=$bh: engine %.1f FPS (call convention %.1f)' % (d['config']['clip_engine']['fps'], d['value']))"
  python tools/bench_configs.py --config 5 --no-cpu-baseline --skip-call-convention --batch-head $bh 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fgfa  batch_head=$bh: %.1f FPS' % d['value'])"
done | tee gpurun_out/c5/batch_head_ab.txt
