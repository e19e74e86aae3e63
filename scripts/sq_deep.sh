#!/bin/bash
# Several SQ counter passes over a short bench run; prints per-wave averages of the step kernel.
# Usage on the GPU box: bash scripts/sq_deep.sh TAG
TAG=${1:-deep}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqd_$TAG; mkdir -p $OUT
CMD="python bench.py --mode ${MODE:-step} --steps 64 --warmup 32 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-profile"
i=0
for SET in "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS" \
           "SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_SENDMSG SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VSKIPPED" \
           "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/run_counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'collide_kernel<true, 1' not in r['Kernel_Name']: continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    w = acc['SQ_WAVES'][0] / max(acc['SQ_WAVES'][1], 1)
    print(f.split('/')[-2], 'waves', w, ' '.join(f"{k}={v[0] / v[1] / max(w, 1):.1f}" for k, v in acc.items() if k != 'SQ_WAVES' and v[1]))
PY
