# round 4, call 10: stem + max-pool in one kernel: bit equality, bench A/B; forced-sharded re-measure after the gather packing
out=gpurun_out/r4c10
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "stem or cat_rows_cast" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $out/pytest_kernels.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default.json 2> $out/bench_default.err
MEGA_STEM_POOL=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_two_kernels.json 2> $out/bench_two_kernels.err
MEGA_FORCE_SHARDED=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_forced_sharded.json 2> $out/bench_forced_sharded.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg > $out/bench_default2.json 2> $out/bench_default2.err
grep -h "timed region:" $out/*.err
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-150
