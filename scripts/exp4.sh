cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in hip ab1 ab2 ab4 ab8 ab15; do
T2D_LIB_NAME=libt2d_$a.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$a -o kt -- python bench.py --config metric --steps 60 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_$a.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/ab_$a/kt_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'][:50]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    if 'collide' in k:
        v=v[len(v)//5:]; print('$a', 'collide avg_us', round(sum(v)/len(v)/1e3,2), 'min', min(v)/1e3)
PY
done
