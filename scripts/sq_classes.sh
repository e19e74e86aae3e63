#!/bin/bash
# Class-resolved dynamic VALU instruction counts of the step kernels (bench.py roofline.issue_cycles_per_class):
# three rocprofv3 PMC passes over a short bench run, per-wave averages of every collide_kernel form that ran.
# Usage on the GPU box: MODE=chain bash scripts/sq_classes.sh TAG   -> gpurun_out/sqc_TAG/classes.json
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqc_$TAG; mkdir -p $OUT
CMD="python bench.py --mode ${MODE:-chain} --steps 64 --warmup 32 --clock-warm 0 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-profile --no-closed-loop"
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('$OUT/p*/run_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'collide_kernel' not in k and 'integrate_kernel' not in k: continue
        k = k[k.find('collide_kernel') if 'collide_kernel' in k else k.find('integrate_kernel'):][:70]
        a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, cs in acc.items():
        w = cs['SQ_WAVES'][0] / max(cs['SQ_WAVES'][1], 1)
        d = out.setdefault(k, {})
        d['waves_per_launch'] = w
        d['launches'] = cs['SQ_WAVES'][1]
        for c, v in cs.items():
            if c != 'SQ_WAVES' and v[1]:
                d[c + '_per_wave'] = v[0] / v[1] / max(w, 1)
json.dump(out, open('$OUT/classes.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
