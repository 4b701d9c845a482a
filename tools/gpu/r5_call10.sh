#!/bin/bash
# round 5, call 10: new tests (unaligned FrozenBN vectors, R-50 in the x3 / wide modes) + smoke with its SP conv check
export TMPDIR=/tmp
out=gpurun_out/r5c10
mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -m gpu -k "unaligned_addresses" > $out/k.log 2>&1; tail -3 $out/k.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu -k "r50_end_to_end or wide_trunk_r50" > $out/e.log 2>&1; tail -4 $out/e.log; grep -E 'R-50 ' $out/e.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
