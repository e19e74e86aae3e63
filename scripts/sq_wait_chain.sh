#!/bin/bash
# The wait / stall side of the step kernel's SQ counters (the passes profile_round.sh leaves out because they slow the kernel:
# ratios are what is read from them).  Usage on the GPU box: MODE=chain bash scripts/sq_wait_chain.sh TAG -> gpurun_out/TAG_sq_wait_<mode>.json
TAG=${1:-r06}
MODE=${MODE:-chain}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqw_${TAG}_$MODE; mkdir -p $OUT
CMD="python bench.py --mode $MODE --steps 416 --warmup 96 --fragment 32 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-closed-loop --no-profile"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && [ $i -gt $PASSES ]; then break; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $SET" >> $OUT/failed.txt
done
python - <<PY
import csv, glob, collections, json, os
frag = 32 if "$MODE" == "chain" else 1
import os
want = os.environ.get("T2D_SQ_FILTER") or ('collide_kernel<true, 1, false, true' if "$MODE" == "chain" else 'collide_kernel<true, 1, false, false')
steps_per_wave = int(os.environ.get("T2D_SQ_STEPS_PER_WAVE", "1"))   # (a LOOP launch: one wave walks through the fragment's steps)
res = {"mode": "$MODE", "kernel_filter": want, "steps_per_launch": frag, "per_wave_and_step": {}, "passes": {}}
for f in sorted(glob.glob('$OUT/p*/run_counter_collection.csv')):
    acc = collections.defaultdict(list)
    names = set()
    for r in csv.DictReader(open(f)):
        if want not in r['Kernel_Name']: continue
        names.add(r['Kernel_Name'][:80])
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    # full fragments only (the run's last launch may be shorter), skip the first quarter
    avg = {k: sum(v[len(v)//4:]) / max(len(v[len(v)//4:]), 1) for k, v in acc.items()}
    w = avg.get('SQ_WAVES', 0.0)
    p = os.path.basename(os.path.dirname(f))
    res["passes"][p] = {"kernel": sorted(names), "launches": len(next(iter(acc.values()))), "per_launch": avg}
    for k, v in avg.items():
        if k != 'SQ_WAVES' and w: res["per_wave_and_step"][k] = v / w / steps_per_wave   # (a chained launch's SQ_WAVES counts every step's waves)
    # kernel durations of this pass (for the slow-down the counters cause)
    kt = glob.glob(os.path.dirname(f) + '/run_kernel_trace.csv')
    if kt:
        d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0])) if want in r['Kernel_Name']]
        d = d[len(d)//4:]
        if d: res["passes"][p]["avg_us_per_step_under_this_pass"] = sum(d) / len(d) / frag
pw = res["per_wave_and_step"]
if 'SQ_WAVE_CYCLES' in pw:
    wc = pw['SQ_WAVE_CYCLES']
    res["shares_of_wave_cycles"] = {k: pw[k] / wc for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM') if k in pw}
json.dump(res, open('gpurun_out/${TAG}_sq_wait_$MODE.json', 'w'), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
rm -rf $OUT/p*/  # raw rocprofv3 directories: gpurun copies <= 64 MiB back
