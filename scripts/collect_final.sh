#!/bin/bash
# copy what scripts/gpu_final.sh TAG left under gpurun_out/ into profiles/ (tracked): summaries, bench lines, attribution passes,
# the tails of the test / soak logs; profiles/traffic_latest.json = this tag's (bench.py reads it, tied to the source hash)
TAG=$1
cd "$(dirname "$0")/.."
for f in summary.json kernel_stats.csv configs.json next_rows.json traffic_latest.json host_path.json bench_driver_like.json \
         bench_1000.json chain_timing_frag20.json sq_wait_chain.json power_clock.json mapgrid_timing.txt; do
    [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
cp gpurun_out/${TAG}_traffic_latest.json profiles/traffic_latest.json
tail -n 3 gpurun_out/${TAG}_logs/pytest_gpu.log > profiles/${TAG}_pytest_gpu_tail.txt
(tail -n 2 gpurun_out/${TAG}_logs/soak.log; tail -n 1 gpurun_out/${TAG}_logs/soak_lidar.log; tail -n 1 gpurun_out/${TAG}_logs/soak_mapgrid.log 2>/dev/null) > profiles/${TAG}_soak_tail.txt
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
s = json.load(open(f"profiles/{tag}_summary.json"))
kt = s["kernel_trace"]
print("source", s["source_sha256"][:16])
for mode in kt:
    for k, v in kt[mode].items():
        if k.startswith("step_kernel") or k.startswith("integrate") or k.startswith("collide_kernel"):
            print(mode, k, round(v["avg_us_per_step"], 3), v["calls"])
for name in ("bench_driver_like", "bench_1000"):
    d = json.loads(open(f"profiles/{tag}_{name}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(name, "%.4g" % d["value"], "us/step %.3f" % (1e3 * d["ms_per_step"]), "frac", r.get("frac"), "at clock", r.get("frac_at_measured_clock"),
          "stale", r.get("counters_stale"), "power", (r.get("power_clock") or {}).get("power_w"))
PY
