mkdir -p gpurun_out/c19
for spb in 7 10; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch $spb > gpurun_out/c19/b_spb$spb.json 2> gpurun_out/c19/b_spb$spb.err
done
grep "timed region" gpurun_out/c19/*.err
