# kernel-trace timeline of the steady state (no overlap): gpurun_out/<tag>/tail.csv = the last 6000 kernels
tag=${1:-trace}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
# blocks of 40 key frames = two step-batches of 20: the stream tail is F(last batch), B(previous batch), B(last batch)
args="--steps 40 --steps-per-batch 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-h2d-leg --no-whole-clip --min-seconds 0.01 --max-blocks 3 --no-overlap"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/t -o t -- python $root/bench.py $args > $root/$out/t.json 2> $root/$out/t.err)
python - $out <<'PY'
import sys,csv,glob
d=sys.argv[1]
f=glob.glob(d+'/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-6000:]
with open(d+'/tail.csv','w') as o:
    w=csv.writer(o)
    for r in rows:
        w.writerow([r['Start_Timestamp'],r['End_Timestamp'],r.get('Queue_Id',''),r.get('Stream_Id',''),r['Kernel_Name'][:110]])
PY
rm -rf $out/t
grep -h "timed region" $out/*.err
