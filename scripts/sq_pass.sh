#!/bin/bash
# SQ counter pass of the default bench (own run, no tracing domains besides kernel-trace)
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o bench -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-profile ${@:2} > $OUT/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$OUT/bench_counter_collection.csv')))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'kernel' in k: print(k[:50], {c: round(sum(v[len(v)//5:]) / len(v[len(v)//5:])) for c, v in d.items()})
PY
