#!/bin/bash
# SQ counter pass of an arbitrary command:  scripts/sq_cmd.sh TAG python scripts/time_integrate.py
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o run -- "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$OUT/run_counter_collection.csv')))
seq = collections.OrderedDict()
for r in rows:
    seq.setdefault((r['Dispatch_Id']), {'k': r['Kernel_Name'].replace('void t2d::(anonymous namespace)::', '')[:40]})[r['Counter_Name']] = float(r['Counter_Value'])
last = None
for d, v in seq.items():
    if 'kernel' not in v['k']: continue
    key = (v['k'], v.get('SQ_WAVES'))
    if key != last:
        w = max(v.get('SQ_WAVES', 1), 1)
        print(v['k'], 'waves', int(w), 'VALU/wave %.0f' % (v.get('SQ_INSTS_VALU', 0) / w), 'SALU/wave %.0f' % (v.get('SQ_INSTS_SALU', 0) / w),
              'active_valu/wave %.0f' % (v.get('SQ_ACTIVE_INST_VALU', 0) / w), 'wave_cycles/wave %.0f' % (v.get('SQ_WAVE_CYCLES', 0) / w),
              'busy_cycles %.0f' % v.get('SQ_BUSY_CYCLES', 0))
        last = key
PY
