mkdir -p gpurun_out/c1; export TMPDIR=/tmp
timeout 60 tools/probes/bin/ldsdma_probe > gpurun_out/c1/probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -rf -s > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1/pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c1/bA.json 2> gpurun_out/c1/bA.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-static-aggregation > gpurun_out/c1/bB.json 2> gpurun_out/c1/bB.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steps-per-batch 5 --no-roofline > gpurun_out/c1/bC.json 2> gpurun_out/c1/bC.err
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c1/bD.json 2> gpurun_out/c1/bD.err
tail -3 gpurun_out/c1/pytest.log; grep "timed region\|host ms" gpurun_out/c1/b*.err
