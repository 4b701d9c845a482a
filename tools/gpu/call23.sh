mkdir -p gpurun_out/c23
for t in 128 256 768; do
MEGA_ATTN_BLOCKS=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c23/b20_$t.json 2> gpurun_out/c23/b20_$t.err
done
MEGA_ATTN_BLOCKS=256 timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c23/b100_256.json 2> gpurun_out/c23/b100_256.err
grep "timed region" gpurun_out/c23/*.err
