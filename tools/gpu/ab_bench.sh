#!/bin/bash
# same-box A/B of the bench: ab_old/ holds an older checkout (git archive <rev> | tar -x -C ab_old, plus the built .so),
# the working tree is B.  usage: bash tools/gpu/ab_bench.sh <tag> [rounds]
tag=${1:-ab}
rounds=${2:-2}
out=$(pwd)/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --min-seconds 3"
for r in $(seq 1 $rounds); do
  (cd $root/ab_old && timeout 200 python bench.py $args > $out/a$r.json 2> $out/a$r.err); echo "A$r $(grep -o 'median [0-9.]*s ([0-9.]* frames/s)' $out/a$r.err)"
  (cd $root && timeout 200 python bench.py $args > $out/b$r.json 2> $out/b$r.err); echo "B$r $(grep -o 'median [0-9.]*s ([0-9.]* frames/s)' $out/b$r.err)"
done
