#!/bin/bash
# gpurun_out/<tag>/ (one tools/profile_round.sh call) -> profiles/<tag>_* : the files the round's README section cites.
#   bash tools/collect_profiles.sh r06
tag=${1:-r06}
src=gpurun_out/$tag
dst=profiles
for f in bench_n1 bench_n1_driver_cli bench_n1_100step_blocks bench_n1_bf16x3 bench_n1_wide_trunk bench_n1_float16 bench_n1_f16x2 \
         bench_n1_forced_sharded bench_n1_igemm2_k128 bench_n1_igemm2_streaming bench_n1_bf16_head_stream bench_n1_unfused_layer1 \
         bench_n1_two_batches_per_block config1 config1_f32 config2 config5; do
  [ -s $src/$f.json ] && cp $src/$f.json $dst/${tag}_$f.json
done
cp $src/ab_legs.txt $dst/${tag}_ab_legs.txt
cp $src/trace_summary.txt $dst/${tag}_trace_summary.txt
cp $src/cli_block_timeline.txt $dst/${tag}_cli_block_timeline.txt
cp $src/bneck_bench.txt $dst/${tag}_bneck_bench.txt
cp $src/pytest_prints.txt $dst/${tag}_parity_prints.txt
(grep -E " passed| failed|pytest rc" $src/pytest_gpu.log | tail -3; tail -2 $src/smoke.log) > $dst/${tag}_pytest_gpu_tail.txt
cp $src/prof/bench_kernel_stats.csv $dst/${tag}_rocprofv3_kernel_stats_bench_n1.csv
cp $src/bench_under_rocprof.json $dst/${tag}_bench_n1_under_rocprofv3.json
python tools/pmc_summary.py $src/pmc_FETCH_SIZE/pmc_counter_collection.csv $src/pmc_WRITE_SIZE/pmc_counter_collection.csv $dst/${tag}_pmc > /dev/null
python tools/pmc_summary.py --mfma-busy $src/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv $src/pmc_GRBM_GUI_ACTIVE/pmc_counter_collection.csv $dst/${tag}_pmc_mfma_busy.csv > /dev/null
for mode in bf16x3:x3 float16:f16 f16x2:f16x2; do
  m=${mode%%:*}; s=${mode##*:}
  cp $src/${m}_prof/bench_kernel_stats.csv $dst/${tag}_${s}_rocprofv3_kernel_stats.csv
  cp $src/${m}_bench_under_rocprof.json $dst/${tag}_${s}_bench_under_rocprofv3.json
  python tools/pmc_summary.py $src/${m}_pmc_FETCH_SIZE/pmc_counter_collection.csv $src/${m}_pmc_WRITE_SIZE/pmc_counter_collection.csv $dst/${tag}_${s}_pmc > /dev/null
  python tools/pmc_summary.py --mfma-busy $src/${m}_pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv $src/${m}_pmc_GRBM_GUI_ACTIVE/pmc_counter_collection.csv $dst/${tag}_${s}_pmc_mfma_busy.csv > /dev/null
done
for m in rdn dff; do [ -s $src/method_$m.json ] && cp $src/method_$m.json $dst/${tag}_method_$m.json; done
# config 5 under rocprofv3 + its per-launch table (round 6)
[ -s $src/c5_prof/c5_kernel_stats.csv ] && cp $src/c5_prof/c5_kernel_stats.csv $dst/${tag}_c5_rocprofv3_kernel_stats.csv
[ -s $src/config5_under_rocprof.json ] && cp $src/config5_under_rocprof.json $dst/${tag}_config5_under_rocprofv3.json
[ -s $src/c5_layers.txt ] && cp $src/c5_layers.txt $dst/${tag}_c5_layers_after.txt
ls $dst | grep "^${tag}_" | wc -l
