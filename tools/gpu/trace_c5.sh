#!/bin/bash
# kernel-trace timeline of config 5's engine in steady state: per queue busy time and the union of busy intervals over the
# last ~40 key frames  ->  gpurun_out/c5/timeline.txt
out=gpurun_out/c5
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/t -o t -- python $root/tools/bench_configs.py --config 5 --no-cpu-baseline --skip-call-convention > $root/$out/t.json 2> $root/$out/t.err)
python - $out <<'PY' | tee $out/timeline.txt
import sys,csv,glob,collections
d=sys.argv[1]
f=glob.glob(d+'/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steady window: the last 6000 kernels before the final ~200 (profiler's family run uses the per-call path at the very end)
rows=[r for r in rows]
# find replays of the warp kernel (one per key frame): use the span covering warp launches -260..-60
warps=[i for i,r in enumerate(rows) if 'fgfa2_kernel' in r['Kernel_Name']]
print("warp launches in trace:", len(warps))
a,b=warps[-140],warps[-60]
seg=rows[a:b]
with open(d+'/window.csv','w') as o:
    for r in seg:
        o.write("%s,%s,%s,%s\n" % (r['Start_Timestamp'], r['End_Timestamp'], r.get('Queue_Id','?'), r['Kernel_Name'].replace(',', ';')[:90]))
nkey=80
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
print("window: %d kernels, %d key frames, span %.3f ms = %.3f ms per key frame" % (len(seg), nkey, (t1-t0)/1e6, (t1-t0)/1e6/nkey))
perq=collections.defaultdict(lambda:[0,0])
iv=[]
for r in seg:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    q=r.get('Queue_Id','?')
    perq[q][0]+=1; perq[q][1]+=e-s
    iv.append((s,e))
for q,(n,t) in sorted(perq.items(), key=lambda kv:-kv[1][1]):
    print("  queue %s: %5d kernels, busy %.3f ms per key frame" % (q, n, t/1e6/nkey))
iv.sort()
u=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: u+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
u+=ce-cs
print("union of busy intervals: %.3f ms per key frame; idle %.3f ms per key frame" % (u/1e6/nkey, ((t1-t0)-u)/1e6/nkey))
fam=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').replace('at::native::','')
    k=n[:60]
    fam[k][0]+=1; fam[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,(n,t) in sorted(fam.items(), key=lambda kv:-kv[1][1])[:28]:
    print("  %-62s n/kf=%5.1f  %.4f ms per key frame" % (k, n/nkey, t/1e6/nkey))
PY
rm -rf $out/t
