# round 4, call 21: attention busy time with copied key sets (new kernel, one segment) vs two segments, same box
MEGA_ATTN_SEGMENTS=0 bash tools/gpu/trace_cli.sh r4c21/copied > /dev/null 2>&1
bash tools/gpu/trace_cli.sh r4c21/segments > /dev/null 2>&1
MEGA_ATTN_SEGMENTS=0 MEGA_ATTN_OCC3=0 bash tools/gpu/trace_cli.sh r4c21/copied_occ2 > /dev/null 2>&1
for m in copied segments copied_occ2; do echo "== $m"; sed -n 1,3p gpurun_out/r4c21/$m/cli_summary.txt; grep "attn_batched\|copy_segments\|CatArrayBatchedCopy<" gpurun_out/r4c21/$m/cli_summary.txt | head -4; done
python - <<'PY'
import csv,re
for mode in ("copied","segments","copied_occ2"):
    rows=list(csv.reader(open('gpurun_out/r4c21/%s/cli_tail.csv'%mode)))
    K=[(int(r[0]),int(r[1]),r[3]) for r in rows]
    pre=[i for i,r in enumerate(K) if 'stem_pool' in r[2]]
    seg=K[pre[-1]:]
    print(mode, [round((r[1]-r[0])/1e3,1) for r in seg if 'attn_batched' in r[2]])
PY
