# round 4, call 20: aggregation as a hipGraph vs eager: per-kernel busy time of the same block on one box
bash tools/gpu/trace_cli.sh r4c20/graph > /dev/null 2>&1
TRACE_ARGS="--aggregation batched-eager" bash tools/gpu/trace_cli.sh r4c20/eager > /dev/null 2>&1
for m in graph eager; do echo "== $m"; sed -n 1,28p gpurun_out/r4c20/$m/cli_summary.txt; done
