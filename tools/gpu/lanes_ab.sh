#!/bin/bash
# box-head graph lanes of the DFF engine, with / without the forked selection inside graph B (same box)
mkdir -p gpurun_out/c5
for nf in 1 0; do for l in 1 2 3 2 4 2; do
  MEGA_NO_FORK_SELECT=$nf python tools/bench_configs.py --method dff --lanes $l 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); e=d['config']['clip_engine']; print('dff no_fork=$nf lanes $l: engine %.1f FPS (call convention %.1f)' % (e['fps'], d['value']))
except Exception as ex:
    print('dff no_fork=$nf lanes $l: CRASHED')"
done; done | tee gpurun_out/c5/lanes_ab.txt
