# round 4, call 14: stem kernels without scratch memory (u32x4_t weight staging): bit equality, kernel time, bench
out=gpurun_out/r4c14
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "stem" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
b="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip"
timeout 300 python bench.py $b > $out/bench_default.json 2> $out/bench_default.err
MEGA_STEM_POOL=0 timeout 300 python bench.py $b > $out/bench_two_kernels.json 2> $out/bench_two_kernels.err
timeout 300 python bench.py $b > $out/bench_default2.json 2> $out/bench_default2.err
for f in $out/bench_*.err; do echo "$(basename $f .err): $(grep -h '\] timed region:' $f | head -1 | cut -c20-150)"; done
export TMPDIR=/tmp; root=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o p -- python $root/bench.py $b --min-seconds 1 > /dev/null 2>&1)
(cd /tmp && MEGA_STEM_POOL=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof2 -o p -- python $root/bench.py $b --min-seconds 1 > /dev/null 2>&1)
grep -h "stem_\|maxpool" $out/prof/p_kernel_stats.csv $out/prof2/p_kernel_stats.csv | cut -c1-200
rm -f $out/prof*/p_kernel_trace.csv
