out=gpurun_out/c25
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --min-seconds 0.01 --max-blocks 3"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/ov -o t -- python $root/bench.py $args > $root/$out/ov.json 2> $root/$out/ov.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/noov -o t -- python $root/bench.py $args --no-overlap > $root/$out/noov.json 2> $root/$out/noov.err)
for d in ov noov; do
python - $out/$d <<'PY'
import sys,csv,glob,gzip
d=sys.argv[1]
f=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-6000:]
with open(d+'/tail.csv','w') as o:
    w=csv.writer(o)
    for r in rows:
        w.writerow([r['Start_Timestamp'],r['End_Timestamp'],r.get('Queue_Id',''),r.get('Stream_Id',''),r['Kernel_Name'][:90]])
PY
rm -f $(find $out/$d -name '*kernel_trace.csv')
done
grep -h "timed region" $out/*.err
ls -la $out/ov $out/noov
