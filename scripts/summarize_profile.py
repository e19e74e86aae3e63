"""Condense the rocprofv3 CSVs of scripts/profile_round.sh into gpurun_out/<tag>_summary.json, <tag>_kernel_stats.csv,
<tag>_configs.json, <tag>_next_rows.json and <tag>_traffic_latest.json (copy to profiles/; bench.py reads
profiles/traffic_latest.json and trusts it only for the sources whose hash it records)."""
import collections, csv, glob, json, os, re, sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tactics2d_amd import build as B


def find(d, suffix):
    g = glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True)
    return g[0] if g else None


def short(name):
    """collide_kernel<WITH_STATUS, FUSE, IOU, CHAIN, LOOP, SPLIT, PIPE>: FUSE >= 0 is the fused step launch; CHAIN / LOOP are the
    two multi-step forms (one workgroup per (env set, step) / resident workgroups looping over the steps), SPLIT one workgroup
    per env, PIPE a loop with integrator waves (1) and lane waves (2); ego_step_kernel<VARIANT, LOOP> likewise"""
    n = name.replace(" ", "")
    m = re.search(r"collide_kernel<([^>]*)>", n)
    if m:
        a = m.group(1).split(",") + ["false"] * 6 + ["0"]
        if a[1] == "-1":
            return "collide_kernel"
        pipe = {"0": "", "false": "", "1": "_pipe", "2": "_pipe2"}.get(a[6], "_pipe")
        return "step_kernel" + ("_chained" if a[3] == "true" else "_loop" if a[4] == "true" else "") + ("_split" if a[5] == "true" else "") + pipe
    m = re.search(r"ego_step_kernel<([^>]*)>", n)   # <VARIANT, LOOP, PIPE>
    if m:
        a = m.group(1).split(",") + ["false", "false"]
        if a[1] == "true":
            return "ego_step_kernel_loop" + ("_pipe" if a[2] == "true" else "")
    for k in ("ego_step_kernel", "lidar_kernel", "idm_kernel", "parking_scene_kernel", "scene_refill_scan_kernel", "scene_refill_kernel",
              "scene_commit_kernel", "derive_kernel", "feedback_policy_kernel", "chain_rollback_kernel", "integrate_kernel",
              "restore_env_kernel", "restore_kernel", "drift_kernel", "frame_pack_kernel"):
        if k in n:
            return k
    return name[:40]


def launch_steps(k, row, fragment):
    """steps one launch of kernel k holds: the grid's y extent for the chained form, the bench's fragment for the loop forms"""
    if k.startswith("step_kernel_chained"):
        return max(1, int(row["Grid_Size_Y"]) // max(1, int(row["Workgroup_Size_Y"])))
    if "_loop" in k:
        return max(1, int(fragment))
    return 1


def steps_of(row, per_step_items):
    g = int(row.get("Grid_Size", 0) or 0)
    return max(1, round(g / per_step_items)) if per_step_items else 1


bench_json = {}
for mode in ("chain", "step"):
    p = os.path.join(out, f"bench_trace_{mode}.log")
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            bench_json[mode] = json.loads(lines[-1])
cfg = (bench_json.get("chain") or bench_json.get("step") or {}).get("config", {})
n_env, agents = cfg.get("envs_per_gpu", 4096), cfg.get("participants_per_env", 64)
log2A = max(1, (agents - 1).bit_length())
epb = 256 >> log2A
per_step_items = (((n_env + epb - 1) // epb + 7) & ~7) * 256    # work-items of one step in a chained launch

summary = {"tag": tag, "source_sha256": B.source_hash(),
           "commands": {m: f"python bench.py --mode {m} --steps 1024 --warmup 128 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-closed-loop" for m in ("chain", "step")}}
stats_csv = []
kt = {}
for mode in ("chain", "step"):
    f = find(f"trace_{mode}", "kernel_trace.csv")
    if f:
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            steps = launch_steps(k, r, cfg.get("fragment", 32))
            d[k].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, steps))
        kt[mode] = {k: dict(calls=len(v), avg_us=sum(x for x, _ in v) / len(v), steps_per_launch=sum(s for _, s in v) / len(v),
                            avg_us_per_step=sum(x for x, _ in v) / sum(s for _, s in v), min_us=min(x for x, _ in v),
                            max_us=max(x for x, _ in v)) for k, v in d.items()}
    f = find(f"trace_{mode}", "kernel_stats.csv")
    if f:
        stats_csv.append(f"# --mode {mode}\n" + open(f).read())
summary["kernel_trace"] = kt


def counters(d, names=None):
    """per kernel: counter -> average per STEP (a chained launch is divided by the steps it holds)"""
    f = find(d, "counter_collection.csv")
    res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if names and r["Counter_Name"] not in names:
                continue
            s = steps_of(r, per_step_items) if k.startswith("step_kernel_chained") else 1
            a = res[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += s
    return {k: {c: v[0] / v[1] for c, v in cs.items() if v[1]} for k, cs in res.items()}


cf, cw = counters("cal_fetch").get("restore_kernel", {}).get("FETCH_SIZE"), counters("cal_write").get("restore_kernel", {}).get("WRITE_SIZE")
N = 4096 * 64
known_r, known_w = 7 * 4 * N, 8 * 4 * N
cal = {}
if cf:
    cal["fetch_kb_reported"] = cf; cal["fetch_factor"] = known_r / (cf * 1024)
if cw:
    cal["write_kb_reported"] = cw; cal["write_factor"] = known_w / (cw * 1024)
summary["traffic_calibration"] = dict(known_read_bytes=known_r, known_write_bytes=known_w, **cal,
                                      note="restore_kernel (mode 0) streams a known byte count with the integrator's 4-B/lane "
                                           "pattern; factor = known / (counter KB * 1024)")
traffic, sq = {}, {}
for mode in ("chain", "step"):
    fe, wr = counters(f"fetch_{mode}"), counters(f"write_{mode}")
    for k in set(fe) | set(wr):
        if "step_kernel" not in k:
            continue
        fb = fe.get(k, {}).get("FETCH_SIZE", 0) * 1024 * cal.get("fetch_factor", 1.0)
        wb = wr.get(k, {}).get("WRITE_SIZE", 0) * 1024 * cal.get("write_factor", 1.0)
        traffic[k] = dict(fetch_bytes_per_step=fb, write_bytes_per_step=wb, hbm_bytes_per_step=fb + wb)
    for k, v in counters(f"sq_{mode}").items():
        if "step_kernel" in k:
            sq[k] = v
    for d in (f"sqc1_{mode}", f"sqc2_{mode}"):   # the class-resolved passes: merged into the kernel's counter set
        for k, v in counters(d).items():
            if "step_kernel" in k and k in sq:
                for c, x in v.items():
                    sq[k].setdefault(c, x)
integ_sq = counters("sq_integ").get("integrate_kernel")
if integ_sq and integ_sq.get("SQ_BUSY_CYCLES"):
    integ_sq["_valu_busy_frac"] = 4.0 * integ_sq.get("SQ_ACTIVE_INST_VALU", 0) / (256 * 4 * integ_sq["SQ_BUSY_CYCLES"] / 32.0)
summary["integrator_sq_per_launch"] = integ_sq
summary["traffic_per_step"] = traffic
summary["sq_counters_per_step"] = sq
summary["bench_json"] = bench_json
for k, v in sq.items():
    if v.get("SQ_WAVES"):
        v["_per_wave"] = {c[3:]: v[c] / v["SQ_WAVES"] for c in v if c.startswith("SQ_INSTS")}
        if v.get("SQ_BUSY_CYCLES"):   # SQ_BUSY_CYCLES is summed over 8 XCDs x 4 SEs; SQ_ACTIVE_INST_* count quad-cycles
            v["_valu_busy_frac"] = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (256 * 4 * v["SQ_BUSY_CYCLES"] / 32.0)
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(summary, open(os.path.join(root, "gpurun_out", f"{tag}_summary.json"), "w"), indent=1)
open(os.path.join(root, "gpurun_out", f"{tag}_kernel_stats.csv"), "w").write("\n".join(stats_csv))

# the other configurations: kernel time per step and per config
cfgs = {}
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    for mode in ("chain", "step"):
        f = find(f"{c}_{mode}", "kernel_trace.csv")
        if not f:
            continue
        d = collections.defaultdict(lambda: [0.0, 0, 0])
        line = [l for l in open(os.path.join(out, f"{c}_{mode}.log")) if l.startswith("{")]
        frag_c = json.loads(line[-1]).get("config", {}).get("fragment", 32) if line else 32
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "kernel" not in k or "rocclr" in k:
                continue
            steps = launch_steps(k, r, frag_c)
            a = d[k]; a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1; a[2] += steps
        cfgs.setdefault(c, {})[mode] = dict(kernels={k: dict(calls=v[1], avg_us=v[0] / v[1], avg_us_per_step=v[0] / v[2]) for k, v in d.items()},
                                            bench_us_per_step=(1e3 * json.loads(line[-1])["ms_per_step"] if line else None))
# ... and the SQ pass of each config's multi-step launches: VALU instructions per SIMD and step, VALU busy
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    f = find(f"{c}_chain_sq", "counter_collection.csv")
    if not f or c not in cfgs:
        continue
    line = [l for l in open(os.path.join(out, f"{c}_chain_sq.log")) if l.startswith("{")]
    frag_c = json.loads(line[-1]).get("config", {}).get("fragment", 32) if line else 32
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "_loop" not in k and "_chained" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[k] += 1
    sqc = {}
    for k, v in acc.items():
        steps = launches[k] * (frag_c if "_loop" in k else 1)   # (chained launches of the small configs are not expected here)
        if not steps or not v.get("SQ_BUSY_CYCLES"):
            continue
        sqc[k] = dict(launches=launches[k], steps_per_launch=frag_c, waves_per_launch=v["SQ_WAVES"] / launches[k],
                      valu_insts_per_simd_and_step=v.get("SQ_INSTS_VALU", 0) / steps / 1024.0,
                      insts_per_simd_and_step=v.get("SQ_INSTS", 0) / steps / 1024.0,
                      valu_busy_frac=4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (256 * 4 * v["SQ_BUSY_CYCLES"] / 32.0))
    cfgs[c]["chain_sq"] = sqc
json.dump(dict(tag=tag, source_sha256=summary["source_sha256"], note="rocprofv3 --kernel-trace of python bench.py --config <cfg> --mode <mode> "
               "--steps 512 --warmup 64: kernel time per launch and per step next to the bench's own wall time per step", configs=cfgs),
          open(os.path.join(root, "gpurun_out", f"{tag}_configs.json"), "w"), indent=1)

# next rows: kernel-trace durations + SQ counters per dispatch
nxt = {"tag": tag, "source_sha256": summary["source_sha256"],
       "note": "rocprofv3 --kernel-trace --stats, then an SQ pass of the same command at 4096 envs; durations in us, counters per dispatch"}
for name, cmd in (("vec", "python scripts/time_vec_env.py 4096"), ("idm", "python scripts/time_idm.py 4096")):
    rows = {}
    f = find(name, "kernel_stats.csv")
    if f:
        for r in csv.DictReader(open(f)):
            rows[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3)
    sqn = {}
    for k, v in counters(name + "_sq").items():
        if v.get("SQ_WAVES") and ("kernel" in k) and "rocclr" not in k:
            sqn[k] = dict(waves=v["SQ_WAVES"], valu_per_wave=v.get("SQ_INSTS_VALU", 0) / v["SQ_WAVES"], salu_per_wave=v.get("SQ_INSTS_SALU", 0) / v["SQ_WAVES"],
                          lds_per_wave=v.get("SQ_INSTS_LDS", 0) / v["SQ_WAVES"], insts_per_wave=v.get("SQ_INSTS", 0) / v["SQ_WAVES"],
                          valu_busy_frac=(4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (256 * 4 * v["SQ_BUSY_CYCLES"] / 32.0) if v.get("SQ_BUSY_CYCLES") else None))
    log = os.path.join(out, name + ".log")
    nxt[name] = dict(command=cmd, kernel_stats=rows, sq=sqn, stdout=open(log).read()[-1200:] if os.path.exists(log) else None)
json.dump(nxt, open(os.path.join(root, "gpurun_out", f"{tag}_next_rows.json"), "w"), indent=1)

# what bench.py reads for roofline.traffic / the VALU-issue roofline (PMC counters cannot be read in-process)
latest = dict(tag=tag, source_sha256=summary["source_sha256"], config=cfg.get("config", "metric"), envs_per_gpu=n_env, participants_per_env=agents,
              hbm_bytes_per_step={k: v["hbm_bytes_per_step"] for k, v in traffic.items()},
              fetch_factor=cal.get("fetch_factor"), write_factor=cal.get("write_factor"),
              source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/profile_round.sh); FETCH_SIZE "
                     "and WRITE_SIZE scaled by the factors calibrated on restore_kernel's known byte count (same 4-B/lane pattern); "
                     "per step: a chained launch's counters are divided by the steps it holds",
              sq_counters_per_step={k: {c: x for c, x in v.items() if not c.startswith("_")} for k, v in sq.items()},
              integrator_sq_per_launch=integ_sq,
              sq_source="rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
                        "SQ_ACTIVE_INST_VALU pass of the headline command in each step mode; SQ_ACTIVE_INST_* count quad-cycles",
              kernel_trace_avg_us_per_step={m: {k: v["avg_us_per_step"] for k, v in d.items()} for m, d in kt.items()})
json.dump(latest, open(os.path.join(root, "gpurun_out", f"{tag}_traffic_latest.json"), "w"), indent=1)
print(json.dumps(dict(kernel_trace=kt, traffic=traffic, sq={k: v.get("_per_wave") for k, v in sq.items()}, cal=cal), indent=1)[:4000])
