mkdir -p gpurun_out/c16
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c16/b20.json 2> gpurun_out/c16/b20.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-ramp > gpurun_out/c16/b20_noramp.json 2> gpurun_out/c16/b20_noramp.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steps-per-batch 12 > gpurun_out/c16/b20_spb12.json 2> gpurun_out/c16/b20_spb12.err
timeout 200 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/c16/b48.json 2> gpurun_out/c16/b48.err
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -rf -k "long_clip or batched or reference_call or inference_loop" > gpurun_out/c16/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c16/pytest.log
grep "timed region\|pre-roll done" gpurun_out/c16/*.err; tail -3 gpurun_out/c16/pytest.log
