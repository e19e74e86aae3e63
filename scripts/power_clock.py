"""Socket power and shader clock sampled THROUGH the headline run (and a few contrast loads): is the chained step's time set by
instructions at a fixed clock, or by a power budget that sets the clock?

    python scripts/power_clock.py [out.json]

A sampler thread reads the SMU's metrics table (amdsmi; sysfs hwmon as a fall-back) as fast as it answers while the main thread
keeps the stream fed.  Per load: us per step (HIP events), power W (mean / max), gfx clock MHz (mean over the XCDs and samples),
the violation / throttle accumulators the firmware keeps (ppt = package power tracking, thermal, ...) before and after.  An
ordinary user cannot move the power cap (rocm-smi --setpoweroverdrive needs root: tried, logged), so the cap experiment the review
asked for is replaced by loads of different energy per instruction at the same occupancy."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_path = sys.argv[1] if len(sys.argv) > 1 else None
import torch  # noqa: E402
import bench as B  # noqa: E402


class Sampler:
    def __init__(self):
        self.h = None
        self.smi = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
        except Exception as exc:   # noqa: BLE001
            self.err = repr(exc)
        self.hwmon = None
        for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            if os.path.exists(os.path.join(d, "power1_average")) or os.path.exists(os.path.join(d, "power1_input")):
                self.hwmon = d
                break
        self.rows = []
        self.stop = False
        self.thread = None

    def one(self):
        r = dict(t=time.perf_counter())
        if self.h is not None:
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                r["power"] = m.get("current_socket_power")
                clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
                if clk:
                    r["gfx"] = float(np.mean(clk))
                    r["gfx_min"] = float(np.min(clk))
                for k in ("average_gfx_activity", "throttle_status", "indep_throttle_status", "accumulation_counter",
                          "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                          "hbm_thm_residency_acc", "temperature_hotspot", "average_socket_power", "current_uclk", "energy_accumulator",
                          "firmware_timestamp", "gfxclk_lock_status"):
                    if k in m and isinstance(m[k], (int, float)):
                        r[k] = m[k]
            except Exception as exc:   # noqa: BLE001
                r["err"] = repr(exc)
        if "power" not in r or not isinstance(r.get("power"), (int, float)) or r.get("power", 0) >= 65535:
            if self.hwmon:
                for f in ("power1_average", "power1_input"):
                    p = os.path.join(self.hwmon, f)
                    if os.path.exists(p):
                        try:
                            r["power"] = int(open(p).read()) / 1e6
                            break
                        except Exception:   # noqa: BLE001
                            pass
                p = os.path.join(self.hwmon, "freq1_input")
                if "gfx" not in r and os.path.exists(p):
                    try:
                        r["gfx"] = int(open(p).read()) / 1e6
                    except Exception:   # noqa: BLE001
                        pass
        return r

    def start(self):
        self.rows, self.stop = [], False

        def loop():
            while not self.stop:
                self.rows.append(self.one())
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()

    def finish(self):
        self.stop = True
        self.thread.join()
        rows = self.rows
        out = dict(samples=len(rows), sample_hz=len(rows) / max(rows[-1]["t"] - rows[0]["t"], 1e-9) if len(rows) > 1 else 0.0)
        # skip the first 30 % of the samples: the clock ramp and the metrics table's own averaging window
        tail = rows[int(0.3 * len(rows)):]
        for k in ("power", "gfx", "gfx_min", "average_gfx_activity", "temperature_hotspot", "current_uclk"):
            v = [r[k] for r in tail if isinstance(r.get(k), (int, float))]
            if v:
                out[k] = dict(mean=float(np.mean(v)), min=float(np.min(v)), max=float(np.max(v)))
        for k in ("accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                  "hbm_thm_residency_acc", "energy_accumulator"):
            v = [r[k] for r in rows if isinstance(r.get(k), (int, float))]
            if len(v) > 1:
                out[k + "_delta"] = v[-1] - v[0]
        ts = [r.get("throttle_status") for r in rows if r.get("throttle_status") is not None]
        if ts:
            out["throttle_status_values"] = sorted(set(int(x) for x in ts))[:8]
        return out


def static_info(s):
    info = {}
    if s.h is not None:
        for name, fn in (("power_cap", "amdsmi_get_power_cap_info"), ("power_info", "amdsmi_get_power_info"),
                         ("violation", "amdsmi_get_violation_status")):
            try:
                v = getattr(s.smi, fn)(s.h)
                info[name] = {k: (x if isinstance(x, (int, float, str)) else str(x)) for k, x in dict(v).items()}
            except Exception as exc:   # noqa: BLE001
                info[name] = repr(exc)
        try:
            info["gfx_clock"] = {k: (x if isinstance(x, (int, float, str)) else str(x))
                                 for k, x in dict(s.smi.amdsmi_get_clock_info(s.h, s.smi.AmdSmiClkType.GFX)).items()}
        except Exception as exc:   # noqa: BLE001
            info["gfx_clock"] = repr(exc)
    for cmd in (["rocm-smi", "--showmaxpower", "--showpower", "--showperflevel"],
                ["rocm-smi", "--setpoweroverdrive", "1000"]):   # (expected to be refused: not root)
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
            info[" ".join(cmd)] = (p.stdout + p.stderr)[-1500:]
        except Exception as exc:   # noqa: BLE001
            info[" ".join(cmd)] = repr(exc)
    return info


def violation(s):
    if s.h is None:
        return None
    try:
        v = dict(s.smi.amdsmi_get_violation_status(s.h))
        return {k: x for k, x in v.items() if isinstance(x, (int, float))}
    except Exception as exc:   # noqa: BLE001
        return repr(exc)


def main():
    dev = torch.device("cuda", 0)
    s = Sampler()
    res = dict(static=static_info(s), loads={})
    seconds = float(os.environ.get("T2D_PC_SECONDS", 4.0))

    def measure(name, enqueue, units_per_call, sync=torch.cuda.synchronize):
        """enqueue() puts ~1-5 ms of work on the stream; repeated for `seconds` with the sampler running"""
        enqueue()
        sync()
        v0 = violation(s)
        s.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        calls = 0
        marks = []
        while time.perf_counter() - t0 < seconds:
            if calls % 64 == 0:   # an event pair per 64 calls over the LAST third = the time per unit at the settled clock
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream() if not hasattr(enqueue, "stream") else enqueue.stream)
                marks.append((calls, ev))
            enqueue()
            calls += 1
            if calls % 16 == 0:   # keep the queue shallow: the sampler's time base is the host's
                sync()
        sync()
        r = s.finish()
        v1 = violation(s)
        k0 = int(len(marks) * 0.6)
        if len(marks) - k0 >= 2:
            (c_a, ev_a), (c_b, ev_b) = marks[k0], marks[-1]
            r["us_per_unit_settled"] = 1e3 * ev_a.elapsed_time(ev_b) / ((c_b - c_a) * units_per_call)
        r["us_per_unit_wall"] = 1e6 * (time.perf_counter() - t0) / (calls * units_per_call)
        if isinstance(v0, dict) and isinstance(v1, dict):
            r["violation_delta"] = {k: v1[k] - v0[k] for k in v1 if k in v0 and k.startswith("acc_") and v1[k] != v0[k]}
            r["violation_after"] = {k: v for k, v in v1.items() if k.startswith("per_") or k.startswith("active_")}
        res["loads"][name] = r
        print(name, json.dumps(r), flush=True)

    # idle
    time.sleep(0.5)
    s.start()
    time.sleep(1.0)
    res["loads"]["idle"] = s.finish()

    frag = 20
    scene = B.build_scene("metric", 4096, 64, seed=0)
    run = B.Runner(scene, dev, "fast")

    def chain():
        run.steps_chain(10 * frag, frag)
    chain.stream = run.stream
    measure("chained_step_metric_4096x64 (unit = step)", chain, 10 * frag)

    def single():
        run.steps_single(64)
    single.stream = run.stream
    measure("one_launch_per_step_metric_4096x64 (unit = step)", single, 64)
    run.pool.set_integrator_variant("exact")
    measure("chained_step_metric_EXACT_integrator (unit = step)", chain, 10 * frag)
    run.pool.set_integrator_variant("fast")
    run.close()
    for n_env in (2048, 1024):
        sc = B.build_scene("metric", n_env, 64, seed=0)
        r2 = B.Runner(sc, dev, "fast")
        r2.pool.set_step_chaining(2, 1)   # the CHAIN form whatever the size (2 / 1 workgroups per CU)

        def chain2(r2=r2):
            r2.steps_chain(10 * frag, frag)
        chain2.stream = r2.stream
        measure(f"chained_step_metric_{n_env}x64 = {n_env // 1024} workgroups per CU (unit = step)", chain2, 10 * frag)
        r2.close()
    # contrast loads: an HBM stream (copy of 1 GiB) and an fp32 GEMM (matrix cores)
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)

    def copy():
        b.copy_(a)
    measure("hbm_copy_1GiB (unit = copy)", copy, 1)
    del a, b
    m = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

    def gemm():
        torch.mm(m, m)
    measure("bf16_gemm_8192 (unit = gemm)", gemm, 1)
    print(json.dumps(res, indent=1))
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
