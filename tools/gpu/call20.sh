mkdir -p gpurun_out/c20
MEGA_FORCE_SHARDED=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c20/sharded.json 2> gpurun_out/c20/sharded.err
echo "rc=$?" >> gpurun_out/c20/sharded.err
tail -5 gpurun_out/c20/sharded.err; cut -c1-200 gpurun_out/c20/sharded.json
