# kernel-trace timeline of ONE driver-CLI block (20 key frames, streams overlapped as in the bench): where does the block's
# wall time go beyond the frame stage?  -> gpurun_out/<tag>/cli_tail.csv + cli_summary.txt
tag=${1:-trace_cli}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-whole-clip --min-seconds 0.01 --max-blocks 4 ${TRACE_ARGS:-}"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/t -o t -- python $root/bench.py $args > $root/$out/t.json 2> $root/$out/t.err)
python - $out <<'PY'
import sys,csv,glob,re
d=sys.argv[1]
f=glob.glob(d+'/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-1500:]
K=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Stream_Id',''),r['Kernel_Name']) for r in rows]
with open(d+'/cli_tail.csv','w') as o:
    w=csv.writer(o)
    for r in K: w.writerow([r[0],r[1],r[2],r[3][:110]])
pre=[i for i,r in enumerate(K) if 'preprocess' in r[3] or 'stem_mfma' in r[3]]      # first kernel of a frame stage
seg=K[pre[-1]:]
t0=seg[0][0]
def nm(n): return re.sub(r'void |\(anonymous namespace\)::|at::native::','',n)[:44]
out=open(d+'/cli_summary.txt','w')
def P(*a):
    s=' '.join(str(x) for x in a); print(s); out.write(s+'\n')
roi=[i for i,r in enumerate(seg) if 'roi_align' in r[3]][-1]
fe=roi+1
if fe+1<len(seg) and 'splitk_finalize' in seg[fe+1][3]: fe+=1
fend=seg[fe][1]
roi=fe-1
P("last block: %d kernels, span %.2f ms; frame stage (preprocess -> first FC) span %.2f ms"%(len(seg),(max(r[1] for r in seg)-t0)/1e6,(fend-t0)/1e6))
busy=sum(r[1]-r[0] for r in seg[:roi+2])
P("frame stage busy %.2f ms"%(busy/1e6))
# idle gaps inside the frame stage
prev=seg[0][1]
for r in seg[1:roi+2]:
    g=r[0]-prev
    if g>15000: P("  frame-stage gap %.1f us before %s at %.2f ms"%(g/1e3,nm(r[3]),(r[0]-t0)/1e6))
    prev=max(prev,r[1])
tail=[r for r in seg[roi+2:]]
# kernels that START after the frame stage ended
late=[r for r in tail if r[0]>=fend]
early=[r for r in tail if r[0]<fend]
P("aggregation kernels overlapping the frame stage: %d (busy %.2f ms); after it: %d, busy %.2f ms, span %.2f ms"%(len(early),sum(r[1]-r[0] for r in early)/1e6,len(late),sum(r[1]-r[0] for r in late)/1e6,((max(r[1] for r in late)-fend)/1e6) if late else 0))
prev=fend; gaps=[]
for r in late:
    g=r[0]-prev
    gaps.append((g,r))
    prev=max(prev,r[1])
P("  gaps after the frame stage: total %.2f ms; > 20 us:"%(sum(max(g,0) for g,_ in gaps)/1e6))
for g,r in gaps:
    if g>20000: P("    %.1f us before %s at +%.2f ms"%(g/1e3,nm(r[3]),(r[0]-fend)/1e6))
from collections import defaultdict
agg=defaultdict(lambda:[0,0])
for r in late:
    agg[nm(r[3])[:36]][0]+=1; agg[nm(r[3])[:36]][1]+=r[1]-r[0]
for n,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:16]: P("    %-38s n=%3d busy %7.1f us"%(n,c,t/1e3))
PY
rm -rf $out/t
grep -h "timed region" $out/*.err
