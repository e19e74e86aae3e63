#!/bin/bash
# instruction counts per wave of the step kernel for several library builds: scripts/sq_variants.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for LIB in "$@"; do
  OUT=gpurun_out/sqv_${LIB%.so}; mkdir -p $OUT
  T2D_LIB_NAME=$LIB timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS --output-format csv -d $OUT -o run -- python scripts/count_step.py > $OUT/log.txt 2>&1
  python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open('$OUT/run_counter_collection.csv')):
    if 'collide_kernel<true, 1' not in r['Kernel_Name']: continue
    a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
w = acc['SQ_WAVES'][0] / max(acc['SQ_WAVES'][1], 1)
print('$LIB', '${T2D_COUNT_CONFIG:-metric}', 'launches', acc['SQ_WAVES'][1], 'waves', w, ' '.join(f"{k[3:]}={v[0] / v[1] / max(w, 1):.0f}" for k, v in sorted(acc.items()) if k != 'SQ_WAVES' and v[1]))
PY
done
