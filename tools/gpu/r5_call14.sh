#!/bin/bash
# round 5, call 14: rocprofv3 kernel stats + PMC passes (one counter per pass) of the split-precision mode as the main configuration
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/r5x3
mkdir -p $out
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o bench -- python $root/bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --min-seconds 1 > $root/$out/bench_under_rocprof.json 2> $root/$out/prof.err); ls $out/prof | head -3
pmcargs="--dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --no-whole-clip --no-graphs --no-overlap --min-seconds 0.01 --max-blocks 1"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/$out/pmc_$c -o pmc -- python $root/bench.py $pmcargs > $root/$out/pmc_$c.json 2> $root/$out/pmc_$c.err); ls -la $out/pmc_$c | tail -1
done
rm -f $out/pmc_*/pmc_kernel_trace.csv $out/prof/bench_kernel_trace.csv
grep -h "timed region" $out/prof.err
