#!/bin/bash
# round 5, call 8: layer2's 128-channel convs on the SP instantiation of igemm8 (MEGA_NARROW_SP=1) against the 128x128 tiles
export TMPDIR=/tmp
out=gpurun_out/r5c8
mkdir -p $out
args="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --no-whole-clip --min-seconds 3"
for r in 1 2; do
  MEGA_NARROW_SP=0 timeout 200 python bench.py $args > $out/a$r.json 2> $out/a$r.err; echo "A$r $(grep -o 'median [0-9.]*s ([0-9.]* frames/s)' $out/a$r.err)"
  MEGA_NARROW_SP=1 timeout 200 python bench.py $args > $out/b$r.json 2> $out/b$r.err; echo "B$r $(grep -o 'median [0-9.]*s ([0-9.]* frames/s)' $out/b$r.err)"
done
