"""Condense the rocprofv3 CSVs of scripts/profile_round.sh into profiles/<tag>_summary.{json,md}."""
import collections, csv, glob, json, os, sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, suffix):
    g = glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True)
    return g[0] if g else None


def short(name):
    if "collide_kernel" in name:   # collide_kernel<WITH_STATUS, FUSE, IOU>: FUSE >= 0 is the fused step launch
        return "collide_kernel" if ",-1" in name.replace(" ", "") else "step_kernel"
    for k in ("integrate_kernel", "restore_env_kernel", "restore_kernel"):
        if k in name:
            return k
    return name[:40]


summary = {"tag": tag, "command": "python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-configs --groups 1"}
f = find("trace", "kernel_trace.csv")
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    summary["kernel_trace"] = {k: dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3,
                                       max_us=max(v) / 1e3, total_ms=sum(v) / 1e6) for k, v in d.items()}
f = find("trace", "kernel_stats.csv")
if f:
    summary["kernel_stats_csv"] = open(f).read()


def counter(d, name):
    f = find(d, "counter_collection.csv")
    res = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                res[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in res.items()}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
cf, cw = counter("cal_fetch", "FETCH_SIZE"), counter("cal_write", "WRITE_SIZE")
N = 4096 * 64
known_r, known_w = 7 * 4 * N, 8 * 4 * N
cal = {}
if "restore_kernel" in cf:
    cal["fetch_kb_reported"] = cf["restore_kernel"]; cal["fetch_factor"] = known_r / (cf["restore_kernel"] * 1024)
if "restore_kernel" in cw:
    cal["write_kb_reported"] = cw["restore_kernel"]; cal["write_factor"] = known_w / (cw["restore_kernel"] * 1024)
summary["traffic_calibration"] = dict(known_read_bytes=known_r, known_write_bytes=known_w, **cal,
                                      note="restore_kernel (mode 0) streams a known byte count with the "
                                           "integrator's 4-B/lane pattern; factor = known / (counter KB * 1024)")
traffic = {}
for k in set(fetch) | set(write):
    fb = fetch.get(k, 0) * 1024 * cal.get("fetch_factor", 1.0)
    wb = write.get(k, 0) * 1024 * cal.get("write_factor", 1.0)
    traffic[k] = dict(fetch_kb_raw=fetch.get(k), write_kb_raw=write.get(k), hbm_bytes_corrected=fb + wb)
summary["traffic"] = traffic
sq = {}
f = find("sq", "counter_collection.csv")
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    sq = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items() if "kernel" in k}
summary["sq_counters_per_dispatch"] = sq
for log in ("bench_trace.log",):
    p = os.path.join(out, log)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            summary["bench_json"] = json.loads(lines[-1])
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(summary, open(os.path.join(root, "gpurun_out", f"{tag}_summary.json"), "w"), indent=1)
if "kernel_stats_csv" in summary:
    open(os.path.join(root, "gpurun_out", f"{tag}_kernel_stats.csv"), "w").write(summary["kernel_stats_csv"])
# what bench.py reads for roofline.traffic / the VALU-issue roofline (PMC counters cannot be read in-process)
bj = summary.get("bench_json", {}).get("config", {})
latest = dict(tag=tag, config=bj.get("config", "metric"), envs_per_gpu=bj.get("envs_per_gpu", 4096),
              participants_per_env=bj.get("participants_per_env", 64), groups=bj.get("env_groups", 1),
              hbm_bytes_per_launch={k: v["hbm_bytes_corrected"] for k, v in traffic.items() if "kernel" in k},
              fetch_factor=cal.get("fetch_factor"), write_factor=cal.get("write_factor"),
              source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/profile_round.sh); FETCH_SIZE "
                     "and WRITE_SIZE scaled by the factors calibrated on restore_kernel's known byte count (same 4-B/lane pattern)",
              sq_counters_per_dispatch=sq,
              sq_source="rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
                        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY pass of the headline command; SQ_ACTIVE_INST_* count quad-cycles",
              kernel_trace_avg_us={k: v["avg_us"] for k, v in summary.get("kernel_trace", {}).items()})
json.dump(latest, open(os.path.join(root, "gpurun_out", f"{tag}_traffic_latest.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel_trace", "traffic_calibration", "traffic") if k in summary}, indent=1)[:3000])
