#!/bin/bash
out=gpurun_out/r3c5
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
grep -n "differ\|tie flips\|has no match" $out/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
