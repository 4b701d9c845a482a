# the whole GPU suite + smoke + the driver-CLI bench line
out=gpurun_out/${1:-r6_full}
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_n1_driver_cli.json 2> $out/bench_n1_driver_cli.err
grep -h "timed region:\|leg\|with H2D" $out/*.err | cut -c1-220
