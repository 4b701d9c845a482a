mkdir -p gpurun_out/c13
timeout 300 python tools/gpu/ablate8.py > gpurun_out/c13/ablate.txt 2>&1
cat gpurun_out/c13/ablate.txt | grep ABL
