#!/bin/bash
# One pass of everything profiles/ needs, on the GPU box (run through gpurun; every step under its own timeout):
#   tools/profile_round.sh <tag> [quick]      e.g. r02  ->  gpurun_out/<tag>/...
# 1. pytest -m gpu   2. __graft_entry__.smoke()   3. bench.py (defaults, incl. CPU baseline unless "quick")
# 4. rocprofv3 kernel stats of the bench   5. rocprofv3 PMC passes, ONE counter per pass, kernel trace only
#    (MI355X_MICROARCH.md HBM section): FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE
#    -> reduce with tools/pmc_summary.py (per kernel family: HBM bytes per launch, MFMA-busy fraction)
tag=${1:-r06}
quick=${2:-}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
if [ -z "$quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -2 $out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
  timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -3 $out/bench_n1.err; cut -c1-260 $out/bench_n1.json
else
  timeout 300 python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; tail -3 $out/bench_n1.err; cut -c1-260 $out/bench_n1.json
fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1_driver_cli.json 2> $out/bench_n1_driver_cli.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o bench -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --min-seconds 1 > $root/$out/bench_under_rocprof.json 2> $root/$out/prof.err); ls $out/prof | head -3
pmcargs="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --no-whole-clip --no-graphs --no-overlap --min-seconds 0.01 --max-blocks 1"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ${EXTRA_PMC:-}; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/$out/pmc_$c -o pmc -- python $root/bench.py $pmcargs > $root/$out/pmc_$c.json 2> $root/$out/pmc_$c.err); ls -la $out/pmc_$c | tail -1
done
# round 6: the other modes as the PROFILED configuration (kernel stats + the same four PMC passes): the split-precision parity
# mode, the fp16 mode, the fp16 two-pass mode  ->  <mode>_prof/, <mode>_pmc_<counter>/ (reduced with tools/pmc_summary.py)
for mode in bf16x3 float16 f16x2; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/${mode}_prof -o bench -- python $root/bench.py --dtype $mode --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --min-seconds 1 > $root/$out/${mode}_bench_under_rocprof.json 2> $root/$out/${mode}_prof.err); ls $out/${mode}_prof | head -2
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/$out/${mode}_pmc_$c -o pmc -- python $root/bench.py --dtype $mode $pmcargs > $root/$out/${mode}_pmc_$c.json 2> $root/$out/${mode}_pmc_$c.err); ls -la $out/${mode}_pmc_$c | tail -1
  done
done
# keep only what the reducers need (the kernel-trace CSVs are large)
rm -f $out/pmc_*/pmc_kernel_trace.csv $out/prof/bench_kernel_trace.csv $out/*_pmc_*/pmc_kernel_trace.csv $out/*_prof/bench_kernel_trace.csv
# the RCCL collectives of the sharded aggregation on ONE GPU (world 1, every key frame owned by rank 0)
leg="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg"
MEGA_FORCE_SHARDED=1 timeout 300 python bench.py $leg > $out/bench_n1_forced_sharded.json 2> $out/bench_n1_forced_sharded.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg > $out/bench_n1_100step_blocks.json 2> $out/bench_n1_100step_blocks.err
# round 5's modes as the main configuration (kernel families of each in the line)
timeout 300 python bench.py --steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_n1_bf16x3.json 2> $out/bench_n1_bf16x3.err
timeout 300 python bench.py --steps 20 --warmup 5 --dtype wide --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_n1_wide_trunk.json 2> $out/bench_n1_wide_trunk.err
# round 6's modes as the main configuration
timeout 300 python bench.py --steps 20 --warmup 5 --dtype float16 --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_n1_float16.json 2> $out/bench_n1_float16.err
timeout 300 python bench.py --steps 20 --warmup 5 --dtype f16x2 --no-cpu-baseline --no-f32-leg --no-h2d-leg --min-seconds 2 > $out/bench_n1_f16x2.json 2> $out/bench_n1_f16x2.err
# round 6's opt-in GEMM kernels end to end (same box): igemm2 on the K <= 128 streaming layers, on the whole streaming class
MEGA_IGEMM2=1 timeout 300 python bench.py $leg > $out/bench_n1_igemm2_k128.json 2> $out/bench_n1_igemm2_k128.err
MEGA_IGEMM2=2 timeout 300 python bench.py $leg > $out/bench_n1_igemm2_streaming.json 2> $out/bench_n1_igemm2_streaming.err
# A/B legs on the same box: the round-3 head (bf16 activation stream), the unfused layer1 blocks, two batches per block
timeout 300 python bench.py $leg --head-stream bfloat16 > $out/bench_n1_bf16_head_stream.json 2> $out/bench_n1_bf16_head_stream.err
MEGA_FUSE_BOTTLENECK=0 timeout 300 python bench.py $leg > $out/bench_n1_unfused_layer1.json 2> $out/bench_n1_unfused_layer1.err
timeout 300 python bench.py $leg --steps-per-batch 10 > $out/bench_n1_two_batches_per_block.json 2> $out/bench_n1_two_batches_per_block.err
timeout 300 python bench.py $leg > $out/bench_n1_driver_cli_again.json 2> $out/bench_n1_driver_cli_again.err
for f in $out/bench_n1*.err; do echo "$(basename $f .err): $(grep -h "\] timed region:" $f | head -1 | cut -c1-150)"; done | tee $out/ab_legs.txt
# kernel timeline of the steady state (per-kernel busy time of one step-batch: tools/trace_summary.py)
bash tools/gpu/trace.sh $tag/trace > /dev/null 2>&1; python tools/trace_summary.py $out/trace/tail.csv > $out/trace_summary.txt 2>&1; head -3 $out/trace_summary.txt
bash tools/gpu/trace_cli.sh $tag/trace_cli > /dev/null 2>&1; cp $out/trace_cli/cli_summary.txt $out/cli_block_timeline.txt; head -6 $out/cli_block_timeline.txt
timeout 120 python tools/gpu/bneck_bench.py > $out/bneck_bench.txt 2>&1; tail -5 $out/bneck_bench.txt
# the other BASELINE configurations
for c in 1 2 5; do timeout 300 python tools/bench_configs.py --config $c > $out/config$c.json 2> $out/config$c.err; cut -c1-160 $out/config$c.json; done
timeout 300 python tools/bench_configs.py --config 1 --dtype float32 --no-cpu-baseline > $out/config1_f32.json 2> $out/config1_f32.err; cut -c1-160 $out/config1_f32.json
for m in rdn dff; do timeout 300 python tools/bench_configs.py --method $m > $out/method_$m.json 2> $out/method_$m.err; cut -c1-120 $out/method_$m.json; done
# config 5 (FGFA) under rocprofv3 (kernel stats of the engine's steady state) + its per-launch table
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/c5_prof -o c5 -- python $root/tools/bench_configs.py --config 5 --no-cpu-baseline > $root/$out/config5_under_rocprof.json 2> $root/$out/c5_prof.err); rm -f $out/c5_prof/c5_kernel_trace.csv; ls $out/c5_prof | head -3
timeout 200 python tools/gpu/flownet_layers.py > $out/c5_layers.txt 2>&1; grep "FlowNetS on" $out/c5_layers.txt
grep -E "ATTRIBUTION|^H |^F bf16|^B bf16|^W wide|^X bf16x3|^bf16x3 |^f16 |^f16x2 |^X3H |^F16H |^H2H |^attention |CALIBRATED|calibrated f32|config [125]|cfg[15] |R-101 600x1000|roi_align bf16|bf16 key frame|graph aggregation:|common-mode|conv2d_sp x3|linear_sp" $out/pytest_gpu.log | cut -c1-460 > $out/pytest_prints.txt
