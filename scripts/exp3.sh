cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in metric cfg3 cfg2; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$c -o kt -- python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/kt_$c.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/kt_$c/kt_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'][:50]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    v=v[len(v)//5:]
    print('$c', k, 'n',len(v),'avg_us', round(sum(v)/len(v)/1e3,2), 'min', min(v)/1e3)
PY
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc2 -o pmc2 -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc3 -o pmc3 -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/pmc3.log 2>&1
python - <<PY
import csv, collections
for f in ('pmc2','pmc3'):
    try: rows=list(csv.DictReader(open(f'gpurun_out/{f}/{f}_counter_collection.csv')))
    except Exception as e: print(f, e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 't2d' not in k: continue
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
