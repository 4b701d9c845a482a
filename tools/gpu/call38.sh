mkdir -p gpurun_out/c38
for t in 256 64; do
MEGA_ATTN_BLOCKS=$t timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c38/b100_$t.json 2> gpurun_out/c38/b100_$t.err
MEGA_ATTN_BLOCKS=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c38/b20_$t.json 2> gpurun_out/c38/b20_$t.err
done
grep "timed region" gpurun_out/c38/*.err
