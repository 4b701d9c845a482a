mkdir -p gpurun_out/c24
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c24/b20.json 2> gpurun_out/c24/b20.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c24/b100.json 2> gpurun_out/c24/b100.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-overlap > gpurun_out/c24/b100_noov.json 2> gpurun_out/c24/b100_noov.err
grep -h "timed region\|host ms" gpurun_out/c24/*.err
