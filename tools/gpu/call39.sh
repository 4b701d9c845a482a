mkdir -p gpurun_out/c39
for i in 1 2 3; do
MEGA_FORCE_SHARDED=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c39/sh$i.json 2> gpurun_out/c39/sh$i.err; echo "rc=$?"
done
grep -h "timed region:" gpurun_out/c39/*.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c39/b20.json 2> gpurun_out/c39/b20.err; grep -h "timed region:" gpurun_out/c39/b20.err
