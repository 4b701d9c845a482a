# last verification of a round on the final code: GPU tests, smoke, the three bench lines
out=gpurun_out/final
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1_driver_cli.json 2> $out/bench_n1_driver_cli.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_n1_100step_blocks.json 2> $out/bench_n1_100step_blocks.err
grep -h "timed region:" $out/*.err
