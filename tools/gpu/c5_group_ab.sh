#!/bin/bash
# same-box A/B of config 5's engine options: key frames per FlowNetS pass (group), one / two streams
mkdir -p gpurun_out/c5
for g in 1 2 3 1 2 4; do
  python tools/bench_configs.py --config 5 --no-cpu-baseline --skip-call-convention --fgfa-group $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $g: %.1f FPS  %.3f ms  blocks %s' % (d['value'], d['ms_per_step'], d['blocks_ms'][:6]))"
done | tee gpurun_out/c5/group_ab.txt
python tools/bench_configs.py --config 5 --no-cpu-baseline --skip-call-convention --fgfa-group 1 --fgfa-no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('group 1, one stream: %.1f FPS' % d['value'])" | tee -a gpurun_out/c5/group_ab.txt
