#!/bin/bash
# same-box A/B of config 5's engine options: key frames per FlowNetS pass (group) with the batched box head
mkdir -p gpurun_out/c5
for g in 4 5 10 20 4; do
  python tools/bench_configs.py --config 5 --no-cpu-baseline --skip-call-convention --fgfa-group $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $g (batched head): %.1f FPS  %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee gpurun_out/c5/group_ab2.txt
