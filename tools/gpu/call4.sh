mkdir -p gpurun_out/c4; export TMPDIR=/tmp
timeout 300 python tools/bench_kernels.py --frames 20 --what attn,pos,roi > gpurun_out/c4/attn.txt 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --min-seconds 0.3 > $GRAFT_REPO_ROOT/gpurun_out/c4/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/c4/prof.err
cd $GRAFT_REPO_ROOT; ls gpurun_out/c4/prof | head; cat gpurun_out/c4/attn.txt | tail -30
