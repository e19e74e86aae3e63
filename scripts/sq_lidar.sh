#!/bin/bash
# SQ counters of the lidar kernel (scripts/time_lidar.py workload)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sq_lidar; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o l -- python scripts/time_lidar.py > $OUT/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$OUT/l_counter_collection.csv')))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[(r['Kernel_Name'][:40], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'lidar' in k[0]: print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
