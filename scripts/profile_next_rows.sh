#!/bin/bash
# rocprofv3 kernel-trace statistics of the "next" rows' kernels (ego step + lidar of VecParkingEnv, IDM, commit / refill of
# the regeneration mode).  Usage on the GPU box: bash scripts/profile_next_rows.sh r02g
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_next_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vec -o vec -- python scripts/time_vec_env.py > $OUT/vec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/idm -o idm -- python scripts/time_idm.py > $OUT/idm.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"tag": tag, "note": "rocprofv3 --kernel-trace --stats; durations in us; the commands time several pool sizes in one "
       "process, so a kernel's row averages over them -- min_us is the small pool, max_us the large one"}
for name, cmd in (("vec", "python scripts/time_vec_env.py"), ("idm", "python scripts/time_idm.py")):
    rows = {}
    for f in glob.glob(os.path.join(out, name, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
                                   max_us=float(r["MaxNs"]) / 1e3, total_ms=float(r["TotalDurationNs"]) / 1e6)
    # per-size averages from the trace itself (grid size tells the pools apart)
    by_grid = {}
    for f in glob.glob(os.path.join(out, name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = by_grid.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
    res[name] = {"command": cmd, "kernel_stats": rows,
                 "by_grid_size": [dict(kernel=k[0][:120], grid_x=k[1], calls=v[0], avg_us=v[1] / v[0]) for k, v in sorted(by_grid.items(), key=lambda kv: -kv[1][1])],
                 "stdout": open(os.path.join(out, name + ".log")).read()[-1500:]}
json.dump(res, open(os.path.join("gpurun_out", f"{tag}_next_rows.json"), "w"), indent=1)
print(json.dumps({k: {n[:60]: round(v["avg_us"], 2) for n, v in res[k]["kernel_stats"].items()} for k in ("vec", "idm")}, indent=1))
PY
