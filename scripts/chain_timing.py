"""Where a step of the CHAINED launch (t2d_step_n, the form `value` is measured on) spends its time, wave by wave.

    T2D_LIB_NAME=libt2d_hip_timing.so T2D_TIMING_WORDS=<words> python scripts/chain_timing.py [frag] [out.json]

(-DT2D_TIMING build.)  Every wave of every step of ONE fragment writes a 32-word record (t2d_collide.hip): cycles per phase, the
cycles it waited for the hand-off, the tail (store drain + barrier + word), HW_ID / XCC_ID, and the constant 100 MHz clock at its
first and last instruction.  A SIMD holds four wave slots; in the steady state a slot serves one wave per step, so

    step time = slot idle between two waves (dispatch) + hand-off wait + start-up + integrate + ... + tail      (per slot)

and that is the table this prints: it sums to the step time.  Beside it: the shader clock the part sustained (cycles / 100 MHz
ticks), how many waves a SIMD held on average, and how many of them were in a phase that can issue arithmetic."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
frag = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out_path = sys.argv[2] if len(sys.argv) > 2 else None
n_env, A = int(os.environ.get("T2D_CT_ENVS", 4096)), 64
padded_wgs = ((n_env // 4) + 7) & ~7
words = frag * padded_wgs * 4 * 32
os.environ.setdefault("T2D_TIMING_WORDS", str(words))
os.environ.setdefault("T2D_LIB_NAME", "libt2d_hip_timing.so")

import torch  # noqa: E402
import bench as B  # noqa: E402
from tactics2d_amd import _ffi  # noqa: E402

dev = torch.device("cuda", 0)
scene = B.build_scene("metric", n_env, A, seed=0)
run = B.Runner(scene, dev, "fast")
assert run.pool.step_form(frag) in ("chain", "step_chain", "chained"), run.pool.step_form(frag)
lib = _ffi.lib()
lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
run.steps_chain(400, frag)     # clocks up, episodes restarting at their usual rate
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(run.stream)
run.steps_chain(10 * frag, frag)
e1.record(run.stream)
torch.cuda.synchronize()
us_per_step_events = 1e3 * e0.elapsed_time(e1) / (10 * frag)
buf = np.zeros(words, np.uint64)
assert lib.t2d_debug_read(run.pool._h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
rec = buf.reshape(frag, padded_wgs, 4, 32)[:, : n_env // 4]     # (the padding workgroups write nothing)
cyc = rec.astype(np.float64)
rt0, rt1 = cyc[..., 16], cyc[..., 17]          # 10 ns ticks
c0, c1 = cyc[..., 15], cyc[..., 20]
life_ticks = rt1 - rt0
clock_ghz = float(((c1 - c0).sum() / life_ticks.sum()) / 10.0)   # cycles per 10 ns -> GHz
span_us = (rt1.max() - rt0.min()) / 100.0
hw = rec[..., 14] & np.uint64(0xffffffff)
xcc = (rec[..., 14] >> np.uint64(32)) & np.uint64(0xf)
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = (hw >> np.uint64(8)) & np.uint64(0xf)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
n_simd = len(np.unique(key))

names = {18: "hand-off wait (poll + barrier)", 0: "start-up: loads, tables + record -> LDS, barrier (a)", 13: "fused integrator",
         1: "pose -> LDS planes, out-of-bound", 2: "sync (b)", 3: "pair broad phase", 4: "pair compaction + SAT", 5: "static box sweep",
         6: "static SAT", 7: "lane box sweep", 8: "lane narrow", 9: "off-lane stage 2 (+ loop exit)", 10: "sync (c)",
         11: "reduce + sync (d)", 12: "status / reward epilogue, auto-reset, stores", 19: "tail: store drain, barrier, word"}
order = [18, 0, 13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 19]
# steady state: drop the fragment's ramp (first 3 steps: every slot starts together) and its drain (last 2)
lo, hi = (3, frag - 2) if frag >= 10 else (0, frag)
S = slice(lo, hi)
steady_steps = hi - lo
# throughput in the steady window, on the constant clock: from the median start of step lo to the median start of step hi
t_start = np.median(rt0.reshape(frag, -1), axis=1)
step_us = float((t_start[hi - 1] - t_start[lo]) / 100.0 / (steady_steps - 1))
life_us = life_ticks[S] / 100.0
to_us = 1.0 / (clock_ghz * 1e3)
phase_us = {k: float(cyc[S][..., k].mean() * to_us) for k in order}
in_kernel = sum(phase_us.values())
slot_idle = step_us - float(life_us.mean())      # a slot serves one wave per step: what is not a wave's life is the slot waiting for one
table = [dict(phase="slot idle between two waves (dispatch of the next workgroup)", us=slot_idle)]
table += [dict(phase=names[k], us=phase_us[k]) for k in order]
table.append(dict(phase="unaccounted inside a wave's life (stamps, first instruction -> first stamp)", us=float(life_us.mean()) - in_kernel))
total = sum(r["us"] for r in table)

# per env kind (wave w of workgroup g holds env 4 g + w; kind = env mod 3: highway / roundabout / intersection)
g_idx = np.arange(n_env // 4)[None, :, None]
w_idx = np.arange(4)[None, None, :]
kind = np.broadcast_to((4 * g_idx + w_idx) % 3, rec.shape[:3])
by_kind = {}
for t, nm in enumerate(("highway", "roundabout", "intersection")):
    m = kind[S] == t
    by_kind[nm] = dict(life_us=float(life_us[m].mean()),
                       phases_us={names[k]: float(cyc[S][..., k][m].mean() * to_us) for k in order})

# residency: waves a SIMD holds on average, and how many of them sit in a phase that issues arithmetic (everything but the
# hand-off wait, the start-up loads and the tail)
t_lo, t_hi = t_start[lo], t_start[hi - 1]
resident = float(np.clip(np.minimum(rt1, t_hi) - np.maximum(rt0, t_lo), 0, None).sum() / ((t_hi - t_lo) * n_simd))
stall_cyc = cyc[..., 18] + cyc[..., 0] + cyc[..., 19]
frac_compute = float(1.0 - stall_cyc[S].sum() / (c1 - c0)[S].sum())
# spread between SIMDs: a SIMD's time per step over the window
per_simd_busy = np.bincount(np.unique(key[S], return_inverse=True)[1].ravel(), weights=life_ticks[S].ravel()) / 100.0 / steady_steps / 4.0
# the fragment's timeline: when the waves of step k start and end, relative to the fragment's first instruction
t00 = rt0.min()
timeline = [dict(step=k, start_p1=float((np.quantile(rt0[k], 0.01) - t00) / 100), start_p50=float((np.median(rt0[k]) - t00) / 100),
                 start_p99=float((np.quantile(rt0[k], 0.99) - t00) / 100), end_p50=float((np.median(rt1[k]) - t00) / 100),
                 end_p99=float((np.quantile(rt1[k], 0.99) - t00) / 100), end_max=float((rt1[k].max() - t00) / 100),
                 life_mean=float(life_ticks[k].mean() / 100), wait_mean=float(cyc[k][..., 18].mean() * to_us),
                 startup_mean=float(cyc[k][..., 0].mean() * to_us))
            for k in range(frag)]
res = dict(
    what="chained step (collide_kernel<true,1,false,CHAIN>), metric scene %d x %d, one fragment of %d steps, steady window = steps %d..%d" % (n_env, A, frag, lo, hi - 1),
    us_per_step_hip_events_10_fragments=us_per_step_events,
    us_per_step_steady_window=step_us,
    fragment_span_us=float(span_us), fragment_span_per_step_us=float(span_us / frag),
    shader_clock_ghz_sustained=clock_ghz,
    simds=n_simd,
    resident_waves_per_simd=resident,
    share_of_a_waves_cycles_in_issuing_phases=frac_compute,
    issuing_waves_per_simd=resident * frac_compute,
    wave_life_us=dict(mean=float(life_us.mean()), p10=float(np.quantile(life_us, 0.1)), p50=float(np.quantile(life_us, 0.5)),
                      p90=float(np.quantile(life_us, 0.9)), max=float(life_us.max())),
    per_simd_mean_wave_life_us=dict(min=float(per_simd_busy.min()), mean=float(per_simd_busy.mean()), max=float(per_simd_busy.max()),
                                    std=float(per_simd_busy.std())),
    attribution_per_slot_and_step=table, attribution_sum_us=total,
    by_env_kind=by_kind,
    timeline_us=timeline,
    note="timing build: the stamps cost ~ +10 % wave cycles (MI355X_MICROARCH.md); read shares, not absolutes, against the product's time",
)
print(json.dumps(res, indent=1))
if out_path:
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
