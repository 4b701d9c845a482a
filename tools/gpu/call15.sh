mkdir -p gpurun_out/c15
for spb in 4 5 10; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch $spb > gpurun_out/c15/b_spb$spb.json 2> gpurun_out/c15/b_spb$spb.err
done
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 10 > gpurun_out/c15/b40_spb10.json 2> gpurun_out/c15/b40_spb10.err
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 10 > gpurun_out/c15/b100_spb10.json 2> gpurun_out/c15/b100_spb10.err
grep "timed region" gpurun_out/c15/*.err
