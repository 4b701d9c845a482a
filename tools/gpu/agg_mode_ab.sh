#!/bin/bash
# same-box A/B at the driver's command line: the batched aggregation replayed from a hipGraph (default) against launched eagerly
mkdir -p gpurun_out/agg
leg="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --min-seconds 3"
for i in 1 2; do
  for mode in batched batched-eager; do
    python bench.py $leg --aggregation $mode > gpurun_out/agg/$mode$i.json 2> gpurun_out/agg/$mode$i.err
    echo "$mode: $(grep -o 'median [0-9.]*s ([0-9.]* frames/s)' gpurun_out/agg/$mode$i.err)"
  done
done | tee gpurun_out/agg/ab.txt
