mkdir -p gpurun_out/c42
MEGA_FORCE_SHARDED=1 timeout 300 python bench.py --steps 20 --warmup 5 --steps-per-batch 20 --no-cpu-baseline --no-roofline > gpurun_out/c42/sh20.json 2> gpurun_out/c42/sh20.err; echo rc=$?
timeout 300 python bench.py --steps 20 --warmup 5 --steps-per-batch 20 --no-cpu-baseline --no-roofline > gpurun_out/c42/b20_spb20.json 2> gpurun_out/c42/b20_spb20.err; echo rc=$?
grep -h "timed region:\|Error\|error" gpurun_out/c42/*.err | head
