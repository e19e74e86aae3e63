# rocprofv3 kernel trace + SQ counters of the grid-tier step (scripts/mapgrid_timing.py, grid tier alone): durations and instructions per wave of map_events_kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export T2D_MG_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mgprof -o mg -- python scripts/mapgrid_timing.py 1024 > gpurun_out/mgprof.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/mgprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/mgpmc -o mg -- python scripts/mapgrid_timing.py 1024 > gpurun_out/mgpmc.log 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob('gpurun_out/mgpmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:50]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in acc.items():
    print(k, n[k], {c: round(x/max(n[k],1),1) for c,x in v.items()})
PY
tail -3 gpurun_out/mgprof.log gpurun_out/mgpmc.log
rm -rf gpurun_out/mgprof gpurun_out/mgpmc
