#!/bin/bash
# One pass of everything profiles/ needs, on the GPU box (run through gpurun; every step under its own timeout):
#   tools/profile_round.sh <tag>      e.g. r01  ->  gpurun_out/<tag>/...
# 1. pytest -m gpu   2. __graft_entry__.smoke()   3. bench.py (defaults)   4. rocprofv3 kernel stats of the bench
# 5. rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, kernel trace only -- MI355X_MICROARCH.md HBM)
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -2 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 400 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -3 $out/bench_n1.err; cut -c1-260 $out/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 48 --warmup 40 --no-cpu-baseline --no-roofline > $out/bench_under_rocprof.json 2> $out/prof.err; ls $out/prof | head -3
# (MFMA-busy: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, again one counter per pass; reduce with
#  `python tools/pmc_summary.py --counter NAME <csv> profiles/<tag>_<NAME>.csv`)
for c in FETCH_SIZE WRITE_SIZE ${EXTRA_PMC:-}; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python bench.py --steps 8 --warmup 26 --no-cpu-baseline --no-roofline --no-graphs --no-overlap > $out/pmc_$c.json 2> $out/pmc_$c.err; ls -la $out/pmc_$c | tail -2
done
