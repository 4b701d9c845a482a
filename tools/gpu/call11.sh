mkdir -p gpurun_out/c11; export TMPDIR=/tmp
root=$(pwd)
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q -rf -k "roi or C_dropins or r101" > gpurun_out/c11/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c11/pytest.log
timeout 100 python tools/bench_kernels.py --frames 20 --what roi > gpurun_out/c11/roi.txt 2>&1
cat > /tmp/one.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
def run(N,H,W,Cin,Cout,R,pad,dil=1):
    x = torch.randn((N,H,W,Cin), device=dev).to(torch.bfloat16); w = (torch.randn((Cout,R,R,Cin), device=dev)*0.05).to(torch.bfloat16)
    sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
    for _ in range(5): ops.conv2d_nhwc(x, w, sc, bi, pad=pad, dil=dil, relu=True)
    torch.cuda.synchronize()
run(20,38,63,256,256,3,1)      # l3.conv2  (igemm8<1>)
run(20,38,63,1024,1024,3,1)    # rpn conv  (igemm8<2>)
run(20,38,63,256,1024,1,0)     # l3.conv3  (igemm8<2>, K=256)
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/c11/pmc_$tag -o pmc -- python /tmp/one.py $root > /dev/null 2> $root/gpurun_out/c11/pmc_$tag.err)
  rm -f gpurun_out/c11/pmc_$tag/pmc_kernel_trace.csv
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c11/bA.json 2> gpurun_out/c11/bA.err
tail -3 gpurun_out/c11/pytest.log; grep "timed region" gpurun_out/c11/b*.err; grep roi_align gpurun_out/c11/roi.txt; ls gpurun_out/c11
