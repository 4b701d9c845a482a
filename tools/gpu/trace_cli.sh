# kernel-trace timeline of ONE driver-CLI block (20 key frames, streams overlapped as in the bench): where does the block's
# wall time go (one or several step-batches per block: later frame stages run beside earlier aggregations)?  -> gpurun_out/<tag>/cli_tail.csv + cli_summary.txt
tag=${1:-trace_cli}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --no-whole-clip --min-seconds 0.01 --max-blocks 4 ${TRACE_ARGS:-}"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/t -o t -- python $root/bench.py $args > $root/$out/t.json 2> $root/$out/t.err)
python - $out <<'PY'
import sys,csv,glob,re,json
from collections import defaultdict
d=sys.argv[1]
f=glob.glob(d+'/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-1500:]
K=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Stream_Id',''),r['Kernel_Name']) for r in rows]
with open(d+'/cli_tail.csv','w') as o:
    w=csv.writer(o)
    for r in K: w.writerow([r[0],r[1],r[2],r[3][:110]])
try:
    nb=len(json.loads(open(d+'/t.json').read().strip().splitlines()[-1])["config"]["batch_sizes_in_a_block"])
except Exception:
    nb=1
def first(n): return 'preprocess' in n or 'stem_mfma' in n or 'stem_pool' in n        # first kernel of a frame stage
pre=[i for i,r in enumerate(K) if first(r[3])]
seg=K[pre[-nb]:]                  # the last timed block: nb step-batches
t0=seg[0][0]
def nm(n): return re.sub(r'void |\(anonymous namespace\)::|at::native::','',n)[:44]
out=open(d+'/cli_summary.txt','w')
def P(*a):
    s=' '.join(str(x) for x in a); print(s); out.write(s+'\n')
# frame-stage kernels = the streams the stem / res5 kernels run on up to each batch's fc0 (the kernel after the last roi_align
# [+ its split-K finalize]); everything else is the aggregation
starts=[i for i,r in enumerate(seg) if first(r[3])]
stages=[]
for b,si in enumerate(starts):
    hi=starts[b+1] if b+1<len(starts) else len(seg)
    roi=[i for i in range(si,hi) if 'roi_align' in seg[i][3]][-1]
    fstream=seg[roi][2]
    fe=[i for i in range(roi+1,hi) if seg[i][2]==fstream][0]
    nxt=[i for i in range(fe+1,hi) if seg[i][2]==fstream]
    if nxt and 'splitk_finalize' in seg[nxt[0]][3]: fe=nxt[0]
    stages.append((seg[si][0],seg[fe][1],fstream,si,fe))
P("last block: %d step-batch(es), %d kernels, span %.2f ms"%(nb,len(seg),(max(r[1] for r in seg)-t0)/1e6))
isF=[False]*len(seg)
sstreams=set()
for (a,e,fs,si,fe) in stages:
    # side streams of the frame stage (res5 beside the RPN branch): streams whose kernels all lie inside [a, e] and carry convs
    for i in range(si,fe+1):
        if seg[i][2]==fs: isF[i]=True
for b,(a,e,fs,si,fe) in enumerate(stages):
    for i in range(si,fe+1):
        if not isF[i] and ('igemm' in seg[i][3] or 'conv64' in seg[i][3] or 'bneck' in seg[i][3]) and seg[i][2]!=fs:
            # a conv on another stream inside the frame stage: res5's side stream, unless that stream is the aggregation's
            sstreams.add(seg[i][2])
agg_streams=set(r[2] for r in seg if 'attn_' in r[3] or 'pos_logits' in r[3])
sstreams-=agg_streams
for b,(a,e,fs,si,fe) in enumerate(stages):
    for i in range(si,fe+1):
        if seg[i][2] in sstreams: isF[i]=True
for b,(a,e,fs,si,fe) in enumerate(stages):
    busy=sum(seg[i][1]-seg[i][0] for i in range(si,fe+1) if isF[i])
    P("frame stage %d: starts +%.2f ms, span %.2f ms, busy %.2f ms"%(b,(a-t0)/1e6,(e-a)/1e6,busy/1e6))
A=[r for i,r in enumerate(seg) if not isF[i]]
fends=[e for (a,e,fs,si,fe) in stages]
def inside(r): return any(a<=r[0]<e for (a,e,fs,si,fe) in stages)
ov=[r for r in A if inside(r)]
late=[r for r in A if r[0]>=fends[-1]]
P("aggregation kernels: %d, busy %.2f ms; started while a frame stage was running: %d (busy %.2f ms); after the last frame stage: %d, busy %.2f ms, span %.2f ms"%(
  len(A),sum(r[1]-r[0] for r in A)/1e6,len(ov),sum(r[1]-r[0] for r in ov)/1e6,len(late),sum(r[1]-r[0] for r in late)/1e6,
  ((max(r[1] for r in late)-fends[-1])/1e6) if late else 0))
prev=fends[-1]; gaps=[]
for r in late:
    gaps.append((r[0]-prev,r)); prev=max(prev,r[1])
P("  gaps after the last frame stage: total %.2f ms; > 20 us:"%(sum(max(g,0) for g,_ in gaps)/1e6))
for g,r in gaps:
    if g>20000: P("    %.1f us before %s at +%.2f ms"%(g/1e3,nm(r[3]),(r[0]-fends[-1])/1e6))
agg=defaultdict(lambda:[0,0])
for r in A:
    agg[nm(r[3])[:36]][0]+=1; agg[nm(r[3])[:36]][1]+=r[1]-r[0]
P("  aggregation kernels of the block by busy time:")
for n,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:16]: P("    %-38s n=%3d busy %7.1f us"%(n,c,t/1e3))
fr=defaultdict(lambda:[0,0])
for i,r in enumerate(seg):
    if isF[i]: fr[nm(r[3])[:36]][0]+=1; fr[nm(r[3])[:36]][1]+=r[1]-r[0]
P("  frame-stage kernels of the block by busy time:")
for n,(c,t) in sorted(fr.items(),key=lambda kv:-kv[1][1])[:14]: P("    %-38s n=%3d busy %7.1f us"%(n,c,t/1e3))
PY
rm -rf $out/t
grep -h "timed region" $out/*.err
