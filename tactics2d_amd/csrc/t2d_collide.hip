// t2d_collide.hip -- event detection for gfx950 (MI355X): participant-vs-participant and
// participant-vs-static-polygon closed-set `intersects`, map-boundary containment, the
// build-defined off-lane test, and the ScenarioManager status / reward epilogue.
//
// Replaces (reference, tactics2d v0.1.9rc3):
//   Vehicle.get_pose          participant/element/vehicle.py:263-281  (bbox order :132-142)
//   Pedestrian.get_pose       participant/element/pedestrian.py:138-149 (centre, radius)
//   StaticCollision.update    traffic/event_detection/collision.py:37-43
//   DynamicCollision.update   traffic/event_detection/collision.py:18-25 (intended semantics)
//   OutBound.update           traffic/event_detection/out_bound.py:37-48
//   OffLane.update            traffic/event_detection/off_lane.py:16-17 (stub; build-defined)
//   _ParkingScenarioManager.check_status + ParkingEnv.step/_get_reward
//                             envs/parking.py:361-392, 243-250, 148-166
//
// Mapping: one workgroup = EPB whole environments (EPB * A_pad <= 256 threads, A_pad =
// max_agents rounded up to a power of two), one lane per participant.  Phases:
//   0. every global load of the workgroup is issued up front so only ONE memory latency is
//      exposed: participant state, map boundary, the 4 shape columns of the type table, and the
//      workgroup's packed static+lane geometry record (fixed stride, built once on the host at
//      t2d_set_*_geometry) copied with 16-B loads straight into dynamic LDS.
//   1. pose: deterministic fp64 sin/cos of the stored heading -> 4 OBB vertices (or circle),
//      written to LDS as coordinate planes s_v[k][lane]; out-of-bound test; conservative fp32
//      box of the pose.
//   2. broad phase -> compacted work queues -> dense narrow phase, three times (participant
//      pairs, static polygons, lane polygons):
//        * pairs: envs that fit a wave (A_pad <= 64) stage (x, y, R) in LDS and compare bounding circles on
//          packed fp32 (two partners per v_pk_* instruction, 1 cm margin), each verdict shifted into the lane's
//          candidate word by v_cmp + v_addc; an env that IS a wave sweeps a ring (lane i against i+1 .. i+32),
//          so each unordered pair is tested once; larger envs walk an LDS spatial-hash grid (cell >=
//          largest circum-diameter, atomicExch-built linked lists).
//        * polygons: a branch-free box sweep over the env's polygons in the LDS record (same packed / addc form).
//      Survivors are compacted (one LDS atomic per lane) into a per-wave LDS queue and processed
//      one entry per lane -- dense lanes instead of per-lane divergent loops, and no register-
//      resident per-participant state, which keeps the kernel at 4 waves / SIMD.  Odd waves visit the
//      polygon stages before the pair stage so that a SIMD's waves are not in the same kind of stretch together.
//      The narrow phase is the oracle's arithmetic: fp64 separating-axis test with strict
//      separation (touching counts, like shapely), results OR-ed into LDS with atomics.
//   3. per-env OR of the flags; one lane per env runs the status epilogue.
//
// Every predicate is the exact arithmetic of oracle/t2d_oracle.c (same operation order,
// -ffp-contract=off, deterministic trig); broad phases are strictly conservative, so flags are
// bit-exact against the oracle's brute force.
// With FUSE >= 0 the same kernel first integrates its participants in registers (t2d_step = one launch).
// Bound: fp64 VALU issue + LDS latency (about 20 B of HBM per participant); see DESIGN.md.
// This translation unit takes the fp64 constants of the deterministic trig kernels from constant memory (t2d_math.h):
// 4 waves per SIMD hide the scalar loads and the kernel is issue-bound (- 1 %); the single-ego kernel, one wave per
// SIMD and latency-bound, keeps them as literals (+ 4 % with the table).
#define T2D_TRIG_TABLE 1
#include "t2d_geom_dev.h"
#include "t2d_idm_dev.h"
#include "t2d_integrate_dev.h"

// s_sleep between two polls of a PIPE progress word, in units of 64 cycles.  A waiting wave shares its SIMD with the wave it
// waits for: measured on cfg3 / cfg4 / cfg5 / cfg2 (us per step) 0: 8.9 / 6.4 / 9.4 / -; 1: 8.8 / 6.4 / 9.2 / 4.9; 4: 8.6 / 6.5 /
// 8.7 / 4.9; 8: 8.6 / 6.6 / 8.6 / 4.9; 16: 8.5 / 6.9 / 8.7; 32: 8.6 / 7.6 / 9.1.
#ifndef T2D_POLL_SLEEP
#define T2D_POLL_SLEEP 4
#endif
namespace t2d {

namespace {

using namespace geom;

// -DT2D_PROBE_SKIP=<bits>: measurement builds that leave a phase out (the flags are then wrong; instruction counts and
// timings of such a build against the full one say what the phase costs): 1 pair stage, 2 pair narrow phase only,
// 4 static polygons, 8 lane polygons, 16 off-lane stage 2 only, 32 pair broad phase only (every pair a candidate: never use),
// 64 the fused integrator (the state stays what it was), 128 the status epilogue, 256 the auto-reset restore
#ifndef T2D_PROBE_SKIP
#define T2D_PROBE_SKIP 0
#endif
// wave priorities of a PIPE launch as three digits: event waves / lane waves / integrator waves (-1: the general rule of the
// step kernel).  Measured on the 1024 x 64 highway / 512 x 32 intersection / 1024 x 64 mixed pools, us per step with lane
// waves: flat 9.98 / 7.62 / 9.21; events first (320) 10.31 / 7.85 / 9.40; integrator first (123) 9.03 / 9.06 / 10.38; the
// general rule 10.14 / 7.83 / 9.38.  The waves of a pair wait for each other: whoever is served first, the other's turn
// comes, and a rule only adds the s_setprio instructions.
#ifndef T2D_PIPE_PRIO
#define T2D_PIPE_PRIO 111
#endif
#ifndef T2D_COLLIDE_WAVES
#define T2D_COLLIDE_WAVES 4  // min waves / SIMD the register allocator must allow
#endif
#ifndef T2D_LOOP_WAVES
#define T2D_LOOP_WAVES 2     // ... of the LOOP forms (a workgroup walks through the steps itself)
#endif
#ifndef T2D_LOOP_RESUM
#define T2D_LOOP_RESUM 0
#endif
#ifndef T2D_LOOP_STAGGER
#define T2D_LOOP_STAGGER 0
#endif
#ifndef T2D_LOOP_PRIO
#define T2D_LOOP_PRIO 0
#endif
#ifndef T2D_CHAIN_PREFETCH
#define T2D_CHAIN_PREFETCH 0   // (measured, same box, ABAB: 16.95 / 16.42 us per step with it, 16.83 / 16.30 without -- not kept on)
#endif
#ifndef T2D_CHAIN_LATE_SPINS
#define T2D_CHAIN_LATE_SPINS 6   // a chained workgroup that had to poll at least this often for its hand-off is LATE: priority 3 throughout (0 = off;
                                 // same box, us per step as fragments of 20 / 100 / single synchronised fragments: off 16.73 / 16.21 / 17.36, 1: 16.63 / 16.21 / 17.23,
                                 // 3: 16.47 / 16.05 / 17.07, 5: 16.41 / 15.99 / 17.01, 8: 16.41 / 15.99 / 17.02 -- profiles/r06_ab_late2.txt)
#endif
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kQueueCap = 288;  // queue entries per wave per round (also holds the broad phase's 3 x 96 staging floats)
constexpr int kMaxHeads = 512;  // EPB * H when the hash grid is in use (A_pad = 128 or 256)
constexpr double kRejectMargin = 1e-6;
constexpr int kLaneShift = 8;   // s_flags bits 8.. hold the off-lane evidence of process_lane

T2D_DEV Quad load_obb_lds(const double* base) {  // &s_v[0][lane], planes kBlock apart
    Quad r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.x[j] = base[(2 * j) * kBlock];
        r.y[j] = base[(2 * j + 1) * kBlock];
    }
    return r;
}

// ---- generic (slow, rare) paths: circles (pedestrians) against polygons streamed from LDS ------
// Kept out of line so their registers do not count against the hot quad-vs-quad code.
struct PolyRef {           // a polygon in LDS: either fp32 interleaved x,y or an OBB in s_v planes
    const float2* f32;     // non-null: fp32 vertices
    const double* planes;  // else: &s_v[0][lane]
    int n;
    T2D_DEV void get(int j, double& x, double& y) const {
        if (f32) {
            const float2 v = f32[j];
            x = (double)v.x;
            y = (double)v.y;
        } else {
            x = planes[(2 * j) * kBlock];
            y = planes[(2 * j + 1) * kBlock];
        }
    }
};

T2D_DEV bool point_in_generic(const PolyRef B, double x, double y) {
    bool in = true;
    double px, py;
    B.get(B.n - 1, px, py);
    for (int j = 0; j < B.n; ++j) {
        double qx, qy;
        B.get(j, qx, qy);
        in &= !(orient(px, py, qx, qy, x, y) < 0.0);
        px = qx;
        py = qy;
    }
    return in;
}

// oracle t2do_circle_convex_intersects
__device__ __noinline__ bool circle_vs_generic(double cx, double cy, double R, const PolyRef B) {
    if (point_in_generic(B, cx, cy)) return true;
    const double R2 = R * R;
    bool hit = false;
    double px, py;
    B.get(B.n - 1, px, py);
    for (int j = 0; j < B.n; ++j) {
        double qx, qy;
        B.get(j, qx, qy);
        hit |= seg_dist2(px, py, qx, qy, cx, cy) <= R2;
        px = qx;
        py = qy;
    }
    return hit;
}

// a_planes: the pose in the LDS coordinate planes (&s_v[0][lane]); b_aos: 4 x (x, y) doubles in global memory
__device__ __noinline__ double quad_iou(const double* a_planes, const double* b_aos) {
    const Quad A = load_obb_lds(a_planes);
    Quad B;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        B.x[k] = b_aos[2 * k];
        B.y[k] = b_aos[2 * k + 1];
    }
    double s[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i + 1) & 3;
        // one term = four independent division chains: enough to keep the issue port busy; the fences stop the
        // scheduler from interleaving all eight terms (33 divisions in flight spill past the kernel's 128 registers)
        s[i] = clipped_edge_term(A.x[i], A.y[i], A.x[k], A.y[k], B, false, A.x[0], A.y[0]);
        __builtin_amdgcn_sched_barrier(0);
        s[4 + i] = clipped_edge_term(B.x[i], B.y[i], B.x[k], B.y[k], A, true, A.x[0], A.y[0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    double inter = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (inter < 0.0) inter = 0.0;
    const double uni = quad_area2(A) + quad_area2(B) - inter;
    return inter / uni;
}

// A store of per-participant output by a launch that holds ONE step (STREAM): the value is not read again inside the launch, and
// what sits dirty in the L2 when the kernel ends is written back behind its last wave -- on the path to the next launch.
// -DT2D_STORE_MODE: 0 plain, 1 nontemporal (streaming; the default), 2 relaxed agent-scope (sc1: written through).  Measured
// at 4096 x 64, one launch per step (scripts/ab_step.py, same box, same checksum): 21.93 / 21.47 / 21.77 us per step; the small
// pools do not care (14.4 / 13.0 / 14.9 either way).  Launches that hold several steps keep plain stores: the next step reads
// them from this XCD's L2.
#ifndef T2D_STORE_MODE
#define T2D_STORE_MODE 1
#endif
template <bool STREAM, class P, class T>
T2D_DEV void st_out(P p, T v) {   // (P: a plain or an address-space-qualified pointer to T)
    if (STREAM && T2D_STORE_MODE == 1) __builtin_nontemporal_store(v, p);
    else if (STREAM && T2D_STORE_MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

T2D_DEV uint32_t cell_hash(int cx, int cy) {
    return ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u);
}

// LDS writes of this wave -> visible to the other lanes of this wave.  The LDS executes one wave's operations in the
// order they were issued, so a read (or atomic) that follows a write in program order sees it whichever lane wrote: all
// that is needed is that the COMPILER keeps the order -- no s_waitcnt (the values a lane uses are waited for where they
// are used, as always), let alone a workgroup-scope release fence, which would also drain every global store in flight.
// (Round 1 had the fences: 1-2 k cycles of store round trip at each sync after the state / flag stores; waiting for
// lgkmcnt(0) at each of the ~25 syncs of a wave still stalled it on every LDS round trip.)
// -DT2D_WAVE_SYNC_WAITCNT builds the conservative form (every LDS operation of the wave retired before the sync):
// tests/test_gpu_soak.py steps both builds through thousands of steps and compares every bit.
T2D_DEV void wave_sync() {
#ifdef T2D_WAVE_SYNC_WAITCNT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
}

// Compact the set bits of every lane's `mask` into the wave's LDS queue (entry = tid | id << 8,
// id = id_base + bit index) and run `process(entry)` densely, one entry per lane; repeats in
// rounds of kQueueCap entries.  Slots are handed out by one LDS atomic per lane on the wave's
// counter (cheaper than a 6-step shuffle scan; the order of entries is irrelevant).  Must be
// called by all 64 lanes of the wave (mask 0 for idle lanes).
template <bool SWAP, bool RING = false, class F>
T2D_DEV void compact_and_process(unsigned long long mask, int id_base, int own_id, uint32_t* queue, int* qcount,
                                 int lane, F process, bool narrow = false, int ring_mask = 63) {
    // entry = participant | other << 8.  SWAP = false: this lane is the participant (own_id = tid)
    // and the bits name the other object (id_base + bit); SWAP = true: this lane owns the other
    // object (own_id = polygon index) and the bits name participants (id_base + bit).
    // narrow (wave-uniform): every lane's mask fits in 32 bits -- the emit loop then runs on half the registers.
    auto entry = [&](int a) -> uint32_t {
        if (RING) {  // bit a = the participant a + 1 places further round its env (pair broad phase, env <= wave)
            const int other = id_base + ((own_id - id_base + 1 + a) & ring_mask);
            return own_id < other ? (uint32_t)own_id | ((uint32_t)other << 8) : (uint32_t)other | ((uint32_t)own_id << 8);
        }
        return SWAP ? (uint32_t)(id_base + a) | ((uint32_t)own_id << 8) : (uint32_t)own_id | ((uint32_t)(id_base + a) << 8);
    };
    for (;;) {
        const int cnt = narrow ? __popc((uint32_t)mask) : __popcll(mask);
        if (__ballot(cnt > 0) == 0ull) break;
        if (lane == 0) *qcount = 0;
        wave_sync();
        const int off = cnt > 0 ? atomicAdd(qcount, cnt) : kQueueCap;
        const int room = kQueueCap - off;
        const int n_emit = room <= 0 ? 0 : (cnt < room ? cnt : room);
        if (narrow) {
            uint32_t m = (uint32_t)mask;
            for (int e = 0; e < n_emit; ++e) {
                const int a = __ffs((int)m) - 1;
                m &= m - 1u;
                queue[off + e] = entry(a);
            }
            mask = m;
        } else {
            for (int e = 0; e < n_emit; ++e) {
                const int a = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                queue[off + e] = entry(a);
            }
        }
        wave_sync();
        const int total = *qcount;
        const int n_round = total < kQueueCap ? total : kQueueCap;
        for (int k = lane; k < n_round; k += 64) process(queue[k]);
        wave_sync();
    }
}

#ifdef T2D_TIMING
#define T2D_MARK(k)                                                                     \
    do {                                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter();                   \
        if (lane == 0) { if (LOOP && !(CHAIN && step_k == step_first)) pv.dbg[wave_slot_ + k] += now_ - t_prev_; else pv.dbg[wave_slot_ + k] = now_ - t_prev_; } \
        t_prev_ = now_;                                                                 \
    } while (0)
#else
#define T2D_MARK(k)
#endif

// collide_kernel<WITH_STATUS, FUSE, IOU, CHAIN, LOOP, SPLIT, PIPE, IDMF>: ONE step body, instantiated in these forms (the flags
// are explained one by one below; `launch_collide` / `launch_step_chain` pick, t2d_step_form names the choice):
//   events only          <*, -1, *>                t2d_collide / t2d_check_status after a separate integrate launch
//   step                 <1, V, IOU>               t2d_step: integrator + events + status in one launch, one wave per env
//   step, split          <1, V, 0, 0, 0, SPLIT>    t2d_step of small pools: one workgroup per env, stages on four waves
//   chained              <1, V, 0, CHAIN>          t2d_step_n of large pools: workgroup (g, k) = step k of env set g
//   chained, split       <1, V, 0, CHAIN, 0, SPLIT>
//   loop                 <1, V, 0, 0, LOOP>        t2d_step_n of pools of <= 2 workgroups per CU: each walks the steps itself
//   loop + PIPE 1 / 2    <1, V, 0, 0, LOOP, 0, P>  ... <= 1 workgroup per CU: integrator waves a step ahead (2: + lane waves)
//   ... + IDMF           <.., IDMF>                step / chained / PIPE 1 with the pool's IDM controllers run inside the launch
// V = 0 / 1: exact / fast integrator.  Every form is held against the plain step launch bit for bit (tests/test_gpu_chain.py).
//
// FUSE = -1: events only (poses read from the pool).  FUSE = 0 / 1: the whole ScenarioManager step in
// one launch -- the participant is first integrated in registers (exact / fast variant, the same
// device functions as integrate_kernel), written back, and its new pose goes straight into the event
// phases: no second launch, no state reload.
// IOU = false: a pool whose status configuration checks neither NoAction nor Arrival (everything but the parking envs)
// runs the instantiation without the quad-IoU code: its out-of-line body brings a 168-B private segment with it.
// CHAIN = true (t2d_step_n): the launch holds several steps.  Workgroup (x, y) takes step y of the envs workgroup x owns and
// is ordered after the workgroup that took their step y - 1 by a per-workgroup word in global memory.  What it buys: no
// launch boundary between steps -- the start-up of step y + 1 and the tail of step y overlap like env groups on separate
// streams do, inside one launch.
// The hand-off stays inside one XCD's L2 (MI355X_MICROARCH.md, inter-workgroup visibility: an agent-scope release / acquire
// pair costs microseconds per workgroup -- the first version of this, with fences, ran 5x slower than separate launches):
//   producer: plain stores -> every wave s_waitcnt vmcnt(0) (acknowledged by the L2) -> barrier -> ONE relaxed agent-scope
//             8-byte store {steps done, XCC id};
//   consumer: one lane polls that word with relaxed agent-scope loads (sc1: served by the L2, never by this CU's L1),
//             barrier, then reads the mutable state with sc1 loads as well.
// That is coherent only if both workgroups run on the same XCD (per-XCD L2s are not coherent with each other).  The grid's x
// extent is a multiple of 8 and the hardware places linear workgroup id i on XCD i mod 8 -- observed, not promised -- so
// the consumer CHECKS it: the producer's XCC id travels in the word, a mismatch (or a wait that runs out: kChainSpinLimit)
// raises chain_err, the host reports the launch as failed and stops chaining.  Never a silent stale read, never a hang.
constexpr int kChainSpinLimit = 1 << 18;   // ~0.2 s of polling: far beyond any step, short enough not to look like a hang
// The kernel's own argument block (PoolView is the first parameter), through an empty asm: loads of its fields through the
// returned pointer cannot be moved above this point.  The compiler otherwise hoists every kernel-argument load it can prove
// invariant to the entry block -- the epilogue's thirty pointers, the geometry layout -- and, out of scalar registers, parks
// them in VGPR lanes for the length of the kernel (v_writelane / v_readlane per value: 108 spilled scalars and ~300 extra
// VALU instructions per wave when the lane stage was rewritten in round 3).
// may this fragment's step 0 store its checkpoint?  Yes unless a failure of an EARLIER fragment is on record (its checkpoint
// is what the host restores): the error word is {code, ckpt_tag of the failing fragment}, written once, as one word
T2D_DEV bool ckpt_open(const KernargView ck) {
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(ck->chain_err), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return w == 0ull || (uint32_t)(w >> 32) == ck->ckpt_tag;
}
T2D_DEV unsigned long long chain_word(uint32_t steps_done) {   // {steps done, XCC id of the workgroup that did the last one}
    return (unsigned long long)steps_done | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg(63508) << 32);
}
// LOOP = true (t2d_step_n on pools of at most two workgroups per CU -- every workgroup resident from the start): the
// workgroup itself walks through the steps, tables and geometry staged once, no hand-off between workgroups at all; its
// register budget is that of two waves per SIMD (256), which a loop round the step body needs: at the 128 registers of
// the metric launch's four waves per SIMD it costs 150 spill slots.  A step of these pools is one wave per
// SIMD walking a dependent chain: the launch boundary, the start-up and the state's round trip through memory are a
// third of it.
// SPLIT = true (pools of at most four waves per SIMD when every env gets a workgroup: <= 4 x CUs envs of 33..64 participants):
// ONE env per workgroup and its event stages on four waves side by side -- wave 0 integrates, then all four derive the poses
// (the same instructions four times over, instead of a hand-over), then wave 0 takes the pairs, wave 1 the static polygons,
// waves 2 and 3 a half of the lane polygons each, and wave 0 reduces and runs the epilogue.  A step of such a pool is one
// wave per SIMD walking a 3300-instruction dependent chain on an otherwise idle SIMD; this way the chain is the integrator
// plus the longest stage (~2000 instructions) and the idle SIMDs do the other stages.  Same arithmetic, same flags.
// PIPE = true (a LOOP launch of at most one workgroup per CU, envs of at most 64 participants): the workgroup carries a second
// set of waves -- wave 4 + w integrates the participants whose events wave w checks -- and the two run a step apart: while
// wave w works through the events of step k, wave 4 + w already integrates step k + 1 from the state it committed for step
// k.  That is speculation on one bit: step k did not end the env's episode.  The event wave publishes that bit with its
// epilogue; an integrator lane whose env did finish restores the snapshot (the restore is its job here: every store to
// the state arrays then comes from one wave, in program order), integrates again from there, and only then commits --
// stores, then the hand-over of (x, y, heading, ids) through LDS.  Nothing of a speculative result is visible anywhere.  A step
// of such a pool was one wave walking integrator + events on an idle SIMD (~9 k + ~15 k cycles on the highway pool); now it
// is the longer of the two plus a hand-shake through two LDS words.  Same arithmetic in the same order per participant.
// PIPE = 2 (pools with lane polygons): a third set of waves -- wave 4 + w takes the lane stage of the participants of wave w,
// which keeps the pairs, the static polygons, the reduce and the epilogue, and waits for the lane wave's bits (OR-ed into
// the same s_flags words) before it reduces; the integrator waves are then waves 8 + w.  Both event waves derive the poses
// from the hand-over, the same instructions twice: the lane stage is the longest event stage of intersection and
// roundabout envs, and on its own wave the events of a step cost about what the integration of the next one does.
constexpr int kPipeSpinLimit = 1 << 17;   // polls (s_sleep 1 between them) before a wait is declared lost: > 10 ms
// IDMF = true (PIPE = 1 launches of pools with installed IDM controllers): the integrator waves run the controller themselves
// ahead of every step -- what t2d_step does with an idm_kernel launch in front of the step launch: the env's positions go
// through the integrator waves' own LDS table, every controlled lane sweeps it for its leader and evaluates the law
// (t2d_idm_dev.h: the kernel's own functions), its acceleration goes to the pool's action field and into the integrator; a
// lane whose env was reset does it again on the restored positions.  Same leaders, same accelerations, same states.
template <bool WITH_STATUS, int FUSE, bool IOU = true, bool CHAIN = false, bool LOOP = false, bool SPLIT = false, int PIPE = 0, bool IDMF = false>
__global__ __launch_bounds__((1 + PIPE) * kBlock, PIPE == 2 ? 3 : (LOOP && !CHAIN) ? T2D_LOOP_WAVES : T2D_COLLIDE_WAVES) void collide_kernel(PoolView pv_arg, t2d_status_config cfg_arg,
                                                                                                                   int interval_ms, int log2A) {
    static_assert(!LOOP || (FUSE >= 0 && WITH_STATUS), "LOOP = the fused step");
    static_assert(!(CHAIN && LOOP) || (!PIPE && !SPLIT && !IDMF && !IOU), "CHAIN + LOOP = the plain chained form whose workgroups take several steps each");
    static_assert(!PIPE || (LOOP && !IOU && !SPLIT), "PIPE = a LOOP launch with integrator waves");
    static_assert(!IDMF || PIPE == 1 || (PIPE == 0 && !LOOP && !SPLIT && FUSE >= 0), "IDMF = the controller inside a PIPE = 1 launch, or ahead of the integrator of a chained one");
    static_assert(!SPLIT || (!LOOP && FUSE >= 0 && WITH_STATUS && !IOU), "SPLIT = the fused step of a plain pool, one launch or chained");
    // `pv` / `cfg` below: the two argument structs -- directly, or (LOOP) through a pointer into the kernel's argument block
    // that is laundered again at the top of every trip, so that what a trip reads of them cannot be hoisted out of the loop:
    // left alone the compiler keeps every invariant argument load live across the whole body (225 spilled scalars, 57
    // spilled vector registers -- at 256 registers).
    auto pvp = [&]() { if constexpr (LOOP) return late_args(); else return &pv_arg; }();
    auto cfgp = [&]() {
        if constexpr (LOOP) return (const __attribute__((address_space(4))) t2d_status_config*)((const __attribute__((address_space(4))) char*)late_args() + kCfgArgOffset);
        else return &cfg_arg;
    }();
#define pv (*pvp)
#define cfg (*cfgp)
    constexpr bool MULTI = CHAIN || LOOP;   // the launch holds several steps: per-step action sets, record slots, sc1 state loads
    __shared__ double s_v[8][kBlock];   // OBB vertex coordinate planes x0,y0,...,x3,y3
    __shared__ float s_cxy[2][kBlock];  // centre x, centre y (the stored fp32 state: exact)
    __shared__ unsigned char s_type[kBlock];  // type id: the radius (circle) / bounding radius (OBB) is read from the table
    // type table in LDS, [column][type]: all columns up to the bounding radius when fused, else only
    // the 4 shape columns (length, width, shape, bounding radius)
    constexpr int kTabCols = T2D_PARAM_COLS;   // (all of them: the last two carry the sub-step and its counts, T2D_P_DT_S / T2D_P_SUBSTEPS)
    __shared__ double s_partab[FUSE >= 0 ? kTabCols * T2D_MAX_TYPES : 4 * T2D_MAX_TYPES];
    __shared__ signed char s_kind[kBlock];  // T2D_SHAPE_* or -1 = inactive
    // event bits OR-ed by the narrow phases (T2D_FLAG_*), and above them (<< kLaneShift) the off-lane evidence of
    // process_lane.  LDS is what decides 4 resident workgroups per CU for the metric scenes: keep this compact.
    __shared__ uint32_t s_flags[kBlock];
    __shared__ uint32_t s_env_or[kBlock / 2];  // per env of the workgroup (an env has at least 2 lanes)
    __shared__ uint32_t s_queue[PIPE == 2 ? 2 * kWaves : kWaves][kQueueCap];   // (PIPE = 2: the lane waves have queues of their own)
    __shared__ int s_qcount[PIPE == 2 ? 2 * kWaves : kWaves];
    // the spatial-hash lists (envs wider than a wave only) live in the queue storage: they are dead
    // before the polygon stages start using the queues (workgroup barrier in between)
    static_assert(kMaxHeads + kBlock <= kWaves * kQueueCap, "hash grid must fit in the queue storage");
    int* const s_head = reinterpret_cast<int*>(&s_queue[0][0]);
    int* const s_next = s_head + kMaxHeads;
    // per env: episode finished.  Shares s_env_or: slot env_local is only ever touched by that env's own
    // lanes, and it is consumed (env_flags written) before it is reused
    int* const s_done = reinterpret_cast<int*>(s_env_or);
    // SPLIT: what wave 0's integrator hands the other three waves -- new x, y, heading and the ids word of every participant
    __shared__ float s_new[SPLIT ? 3 : 1][SPLIT ? 64 : 1];
    __shared__ uint32_t s_ids_new[SPLIT ? 64 : 1];
    // PIPE: the integrator wave's hand-over (x, y, heading, ids per participant), the event wave's verdict per env (episode
    // over), and the two progress words of every wave pair: steps committed / steps decided
    // (the hand-over has two buffers, by step parity: the integrator fills the next step's while it still waits for this
    // step's verdict -- a speculative result too, but nobody reads a buffer before the progress word says it is committed)
    __shared__ float s_hand[PIPE ? 2 : 1][PIPE ? 3 : 1][PIPE ? kBlock : 1];
    __shared__ uint32_t s_hand_ids[PIPE ? 2 : 1][PIPE ? kBlock : 1];
    __shared__ uint32_t s_dec[PIPE ? kBlock / 2 : 1];
    __shared__ uint32_t s_seq_i[PIPE ? kWaves : 1], s_seq_e[PIPE ? kWaves : 1], s_seq_b[PIPE == 2 ? kWaves : 1];
    // IDMF: the integrator waves' table of their envs' positions (NaN = inactive slot) and speeds, as in idm_kernel
    __shared__ double2 s_ixy[IDMF && PIPE ? kBlock : 1];
    __shared__ float s_iv[IDMF && PIPE ? kBlock : 1];
    // CHAIN: the workgroup found its predecessor unfinished -- its env set is on the fragment's critical path (see chain_wait)
    __shared__ int s_late[1];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_geo[];  // packed geometry record

    // Every kernel argument the start-up phase needs, requested in ONE scalar round trip.  Left to itself the compiler
    // fetches each PoolView field where it is first used -- behind `if (valid)`, behind the previous batch's wait -- and
    // the phase becomes a chain of five or six dependent s_load round trips, each a scalar-cache miss while all 4096
    // waves of the launch start together: 3.7 k of the phase's 5.3 k cycles (-DT2D_TIMING) before the first state load
    // is even issued.  The empty asm makes all of them live here, so their loads are issued back to back.
    // (as_global: a plain pointer comes out of the asm generic, and generic means flat_load, which also counts against
    // the LDS wait counter -- every LDS wait would then wait for the state loads in flight as well)
    auto a_ids = as_global(pv.ids);
    auto a_x = as_global(pv.x), a_y = as_global(pv.y), a_h = as_global(pv.heading), a_v = as_global(pv.speed);
    auto a_act0 = as_global(pv.act0), a_act1 = as_global(pv.act1);
    auto a_idm = as_global(pv.idm_ctrl);
    auto a_params = as_global(pv.params);
    auto a_geo = as_global(pv.geo);
    int a_n_env = pv.n_env, a_A = pv.A, a_stride = pv.geo_layout.stride, a_epb = pv.geo_layout.epb, a_act_stride = pv.act_stride;
    // (in-out operands: the values after the asm are new to the compiler, so it keeps them in registers instead of
    // dropping them and fetching the same arguments again behind the next branch)
    asm volatile("" : "+s"(a_ids), "+s"(a_x), "+s"(a_y), "+s"(a_h), "+s"(a_v), "+s"(a_act0), "+s"(a_act1), "+s"(a_idm),
                      "+s"(a_params), "+s"(a_geo), "+s"(a_n_env), "+s"(a_A), "+s"(a_stride), "+s"(a_epb), "+s"(a_act_stride));
    // Placement of the step launch (t2d_debug_set_step_placement): which logical workgroup -- which EPB envs -- this physical
    // workgroup steps, and by how many waves its lane -> participant map is rotated.  Results do not depend on it; the
    // hardware places workgroup b on XCD b mod 8 and its waves on fixed SIMDs, so the map decides which envs share a SIMD.
    // what this workgroup stands for: a set of EPB envs = one geometry record -- or (SPLIT) ONE env, whose record it shares
    const int unit = blockIdx.x;
    int wg = SPLIT ? unit / a_epb : unit, wave_rot = 0;
    // CHAIN + LOOP: workgroup (x, y) takes the steps [y k, (y + 1) k) of its envs, k = pv.loop_steps -- the dispatch gap, the
    // hand-off, the start-up (tables and record staged once per workgroup) and the store drain are paid once per k steps
    const int steps_here = (CHAIN && LOOP) ? pv.loop_steps : 1;
    int step_k = CHAIN ? (int)blockIdx.y * steps_here : 0;
    [[maybe_unused]] const int step_first = step_k;
    [[maybe_unused]] const int step_end = CHAIN ? step_k + steps_here : (LOOP ? pv.loop_steps : 1);
    if (CHAIN && unit >= pv.chain_real_wgs) {   // padding of the grid's x extent to a multiple of 8 (see launch_step_chain)
        if (threadIdx.x == 0)
            __hip_atomic_store(&pv.chain_done[unit], chain_word(pv.chain_base + (uint32_t)(step_k + steps_here)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#ifdef T2D_DEBUG_HOOKS   // (libt2d_hip_debug.so: t2d_debug_set_step_placement)
    if (!CHAIN && !SPLIT && pv.wgmap) {
        const uint32_t m = pv.wgmap[blockIdx.x];
        wg = (int)(m & 0xffffu);
        wave_rot = (int)(m >> 16);
    }
#endif
    // `tid` = the participant's slot in the workgroup's LDS tables.  SPLIT: the four waves all stand for the env's 64 slots;
    // `role` tells them apart and `ptid` is the thread's own number (what the staging loops stride by)
    const int role = SPLIT ? (int)(threadIdx.x >> 6) : 0;
    const int ptid = (int)threadIdx.x;
    // PIPE = 2: threads [E, 2E) of the block (E = EPB << log2A) are the lane waves: same slots, same participants as [0, E)
    const bool role_b = PIPE == 2 && (int)threadIdx.x >= (a_epb << log2A);
    const int tid = SPLIT ? (int)(threadIdx.x & 63u)
                  : PIPE == 2 ? (int)threadIdx.x - (role_b ? (a_epb << log2A) : 0)
                          : (blockDim.x == kBlock ? (int)((threadIdx.x + 64u * (unsigned)wave_rot) & (kBlock - 1u)) : (int)threadIdx.x);
    [[maybe_unused]] const int lane = tid & 63;   // (the step body derives its own lane coordinates: see the loop below)
    const int A_pad = 1 << log2A;
    const int EPB = a_epb;
    const int nthreads = SPLIT ? kBlock : EPB << log2A;   // (SPLIT: four waves per env whatever the record holds)
    const int env_local = SPLIT ? unit % EPB : tid >> log2A;   // the env's place in its geometry record
    const int agent = tid & (A_pad - 1);
    const int env = SPLIT ? unit : wg * EPB + env_local;
    const bool valid = env < a_n_env && agent < a_A;
    [[maybe_unused]] const int idx = valid ? env * a_A + agent : 0;
    const bool use_hash_grid = log2A > 6;  // envs larger than a wave use the LDS spatial hash
    const int H = 2 * A_pad;               // buckets per env (power of two)
    [[maybe_unused]] uint32_t* const queue = s_queue[SPLIT ? role : (tid >> 6) + (role_b ? kWaves : 0)];
    [[maybe_unused]] int* const qcount = &s_qcount[SPLIT ? role : (tid >> 6) + (role_b ? kWaves : 0)];

#ifdef T2D_TIMING
    unsigned long long t_prev_ = __builtin_readcyclecounter();
    // CHAIN: one record of 32 words per wave AND step (scripts/chain_timing.py): words 0-15 as below, 16 / 17 = the constant
    // 100 MHz clock (s_memrealtime) at the wave's start / end, 18 = cycles waiting for the hand-off, 19 = cycles of the tail
    // (store drain + barrier + the word), 20 = the cycle counter at the end
    const size_t wave_slot_ = CHAIN ? ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6)) + (threadIdx.x >> 6)) * 32
                                    : ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    if (lane == 0) {  // where and when this wave ran: HW_ID | XCC_ID << 32, start tick
        pv.dbg[wave_slot_ + 14] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                                  ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
        pv.dbg[wave_slot_ + 15] = t_prev_;
        if (CHAIN) pv.dbg[wave_slot_ + 16] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // Wave priority = how far BEHIND a wave is: 3 through the integrator and the pose phase, 2 in the first event stage,
    // 1 in the second, 0 from the reduce on.  The launch is one wave-round (four waves per SIMD that start together and
    // are never replaced), so a SIMD is busiest while all four are alive: with the arbiter serving the wave with the
    // least progress first they finish close together, instead of one after the other with the last one running alone at
    // half the issue rate.  Measured: 27.45 us per step without priorities (27.4 with the round-1 rule "event phases
    // first"), 26.1 with two levels, 25.0 with these four (3 / 3 / 2 / 0 and 3 / 2 / 2 / 0: 25.6 / 26.0).
    // When the launches of several env groups overlap (pv.overlapped, t2d_step_groups) the opposite holds: a workgroup
    // that retires makes room for the next launch's, so waves past the integrator go first (0, then 2) -- 4 groups:
    // 20.0 us per step of all envs with that rule, 20.3 without priorities, 23.4 with the single-launch rule.
    // LOOP without integrator waves on pools of more than two workgroups per CU (T2D_LOOP_PRIO = 1): a SIMD's four waves never
    // retire, so "priority, then age" would let its oldest wave run ahead of the others for the whole fragment and leave the
    // youngest to finish alone.  The priority ROTATES instead -- (step + dispatch round of the workgroup) mod 4: whenever the
    // four are in the same step their priorities are a permutation, and over four steps each has held every level
    // (T2D_LOOP_PRIO = 2: the single launch's phase rule, re-armed on every trip)
    constexpr bool loop_rot = LOOP && !PIPE && T2D_LOOP_PRIO == 1;
    constexpr bool pipe_prio = (PIPE != 0 && T2D_PIPE_PRIO >= 0) || loop_rot;
    const bool behind_first = (pv.overlapped == 0 || (LOOP && !PIPE && T2D_LOOP_PRIO == 2)) && !pipe_prio;
    if (behind_first) __builtin_amdgcn_s_setprio(3);
    if constexpr (pipe_prio && !loop_rot) {
        if ((int)threadIdx.x >= PIPE * (a_epb << log2A)) __builtin_amdgcn_s_setprio(T2D_PIPE_PRIO >= 0 ? T2D_PIPE_PRIO % 10 : 0);
        else if (role_b) __builtin_amdgcn_s_setprio(T2D_PIPE_PRIO >= 0 ? (T2D_PIPE_PRIO / 10) % 10 : 0);
        else __builtin_amdgcn_s_setprio(T2D_PIPE_PRIO >= 0 ? (T2D_PIPE_PRIO / 100) % 10 : 0);
    }
    // chained launch: wait for the step before this one of the same envs (one lane polls, s_sleep between polls).
    // A hand-off that fails -- the wait ran out, or the producer sat on another XCD -- is RECORDED, {what, which fragment}, and
    // the workgroup goes on (never a hang): the host rolls the pool back to the checkpoint the fragment wrote in its first
    // step (below) and reports the launch as failed, so nothing computed from here on is ever handed to a caller.
    [[maybe_unused]] auto chain_fail = [&](uint32_t code) {   // (the first failure names the fragment; {code, tag} is ONE word:
        // a reader never sees the code of one failure with the tag of another, or a code whose tag has not landed yet)
        unsigned long long none = 0ull;
        (void)__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long*>(pv.chain_err), &none,
                                                   (unsigned long long)code | ((unsigned long long)pv.ckpt_tag << 32),
                                                   __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    [[maybe_unused]] bool chain_late = false;
    auto chain_wait = [&]() {
        if (ptid == 0) {
            const uint32_t want = pv.chain_base + (uint32_t)step_k;
            int spins = 0;
            unsigned long long w;
            // (a signed difference: the counter may wrap after 2^32 steps of a pool)
            while ((int32_t)((uint32_t)(w = __hip_atomic_load(&pv.chain_done[unit], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - want) < 0) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                // give up when the wait ran out -- or when another workgroup's did (checked now and then): one failure
                // ends the launch in about one limit, not in one limit per workgroup
                if (spins > kChainSpinLimit ||
                    ((spins & 255) == 0 && __hip_atomic_load(pv.chain_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    chain_fail(1u);
                    w = chain_word(want);
                    break;
                }
            }
            // the previous step's stores sit in ITS XCD's L2: the sc1 loads below see them only from the same XCD
            if ((uint32_t)(w >> 32) != (uint32_t)__builtin_amdgcn_s_getreg(63508)) chain_fail(2u);
            s_late[0] = T2D_CHAIN_LATE_SPINS > 0 && spins >= T2D_CHAIN_LATE_SPINS;
        }
        __syncthreads();
        // A fragment lasts as long as its SLOWEST chain of workgroups (scripts/chain_timing.py, timeline: the last env set ends
        // 28 us after the median one in a 20-step fragment): a workgroup that had to wait for its predecessor belongs to such a
        // chain -- its waves take priority 3 for the whole step, ahead of the waves of env sets that are on time.
        if (T2D_CHAIN_LATE_SPINS > 0 && s_late[0]) {
            chain_late = true;
            __builtin_amdgcn_s_setprio(3);
        }
    };
#if T2D_LOOP_STAGGER > 0   // (experiment: start the workgroups of dispatch round k -- the k-th on their CU -- k x T2D_LOOP_STAGGER ticks of
    // the 100 MHz clock late, so that a SIMD's resident waves sit in different phases of the step)
    if constexpr (LOOP && !PIPE) {
        const long long wait = (long long)(blockIdx.x >> 8) * T2D_LOOP_STAGGER;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    if constexpr (LOOP) {
        // tables and the workgroup's geometry record: staged once, ahead of the loop, in the plainest form (kept inside
        // the loop behind a first-trip test, the staging code's lane masks were hoisted out of it and held in scalar
        // registers for the whole body: 195 spilled scalars)
        constexpr int kTab = kTabCols * T2D_MAX_TYPES;
        const int nthr = (int)blockDim.x;
        for (int q = (int)threadIdx.x; q < kTab; q += nthr) s_partab[q] = a_params[q];
        if (a_geo) {
            typedef uint32_t u32x4l __attribute__((ext_vector_type(4)));
            const T2D_GLOBAL u32x4l* src = (const T2D_GLOBAL u32x4l*)(a_geo + (size_t)wg * a_stride);
            for (int q = (int)threadIdx.x; q < (a_stride >> 2); q += nthr) reinterpret_cast<u32x4l*>(s_geo)[q] = src[q];
        }
    }
    // a bounded wait on one of the pair's progress words (every lane reads the same LDS word); a wait that runs out raises
    // chain_err -- the host then reports the launch as failed -- and the wave goes on: never a hang
    [[maybe_unused]] auto pipe_wait = [&](uint32_t* word, uint32_t want) {
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
            __builtin_amdgcn_s_sleep(T2D_POLL_SLEEP);
            if (++spins > kPipeSpinLimit) {
                __hip_atomic_store(pv.chain_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        asm volatile("" ::: "memory");   // what the other wave wrote before it moved the word is read after this point
    };
    [[maybe_unused]] auto pipe_post = [&](uint32_t* word, uint32_t value) {
        // (a wave's LDS operations complete in order: the plain writes above are in the LDS before the word moves; the
        // conservative build -- tests/test_gpu_soak.py -- waits for them first, and for what it read, like wave_sync)
#ifdef T2D_WAVE_SYNC_WAITCNT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("" ::: "memory");
#endif
        if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if constexpr (PIPE) {
        if (threadIdx.x < (unsigned)kWaves) {
            s_seq_i[threadIdx.x] = s_seq_e[threadIdx.x] = 0u;
            if (PIPE == 2) s_seq_b[threadIdx.x] = 0u;
        }
        if (threadIdx.x < (unsigned)(kBlock / 2)) s_dec[threadIdx.x] = 0u;
        if ((int)threadIdx.x >= PIPE * nthreads) {
            // ======== integrator waves: thread PIPE * nthreads + t serves the participant of event thread t ========
            const int t = (int)threadIdx.x - PIPE * nthreads;
            const int w = t >> 6;
            const int i_env_local = t >> log2A;
            const int i_env = wg * EPB + i_env_local;
            const bool i_valid = i_env < a_n_env && (t & (A_pad - 1)) < a_A;
            const int i_idx = i_valid ? i_env * a_A + (t & (A_pad - 1)) : 0;
            __syncthreads();   // (a) of the event waves' first trip: tables staged, progress words cleared
            uint32_t ids = 0;
            float x = 0, y = 0, h = 0, v = 0, vx = 0, vy = 0;
            if (i_valid) {
                ids = ld_state<true>(a_ids + i_idx);
                x = ld_state<true>(a_x + i_idx);
                y = ld_state<true>(a_y + i_idx);
                h = ld_state<true>(a_h + i_idx);
                v = ld_state<true>(a_v + i_idx);
                if (((ids >> kIdsModelShift) & 0xff) == T2D_MODEL_POINTMASS) {
                    vx = ld_state<true>(as_global(pv.vx) + i_idx);
                    vy = ld_state<true>(as_global(pv.vy) + i_idx);
                }
            }
            // IDMF: the lane's controller (fixed for the launch) and its row
            [[maybe_unused]] bool has_ctrl = false;
            [[maybe_unused]] idm::IdmRow crow{};
            if constexpr (IDMF) {
                const int ctrl = i_valid ? (int)as_global(pv.idm_ctrl_all)[i_idx] : T2D_IDM_NONE;
                has_ctrl = ctrl != T2D_IDM_NONE && ctrl < pv.idm_n_ctrl;
                if (has_ctrl) crow = idm::load_row(as_global(pv.idm_rows) + (size_t)ctrl * T2D_IDM_COLS);
            }
            const int n_steps = pv.loop_steps;
            const size_t act_step = (size_t)pv.chain_act_step;
            const size_t ai0 = (size_t)i_idx * a_act_stride;
            float a0 = 0, a1 = 0;
            if (i_valid) {
                a0 = a_act0[ai0];
                a1 = a_act1[ai0];
            }
#ifdef T2D_TIMING
            unsigned long long it_prev_ = __builtin_readcyclecounter();
            const size_t i_slot_ = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
#define T2D_IMARK(q)                                                                  \
    do {                                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter();                 \
        if ((threadIdx.x & 63u) == 0u) pv.dbg[i_slot_ + q] += now_ - it_prev_;        \
        it_prev_ = now_;                                                              \
    } while (0)
#else
#define T2D_IMARK(q)
#endif
            for (int k = 0; k <= n_steps; ++k) {
                const KernargView ia = late_args();   // (per trip: see the event waves' loop)
                float na0 = 0, na1 = 0;               // the next step's actions: their latency overlaps this integration
                if (i_valid && k + 1 < n_steps) {
                    na0 = a_act0[ai0 + (size_t)(k + 1) * act_step];
                    na1 = a_act1[ai0 + (size_t)(k + 1) * act_step];
                }
                // what this step makes of the lane's state (stays the state itself for lanes the integrator skips)
                float nx = x, ny = y, nh = h, nv = v, nvx = vx, nvy = vy, app0 = 0, app1 = 0;
                bool moved = false, has_vel = false;
                bool todo = k < n_steps;         // lanes to integrate in this round
                bool decided = k == 0;           // the previous step's verdict is in (nothing precedes step 0)
                for (;;) {                       // at most two rounds: the speculative one, and one for envs that were reset
                    const int model = (ids >> kIdsModelShift) & 0xff;
                    const int type = (ids >> kIdsTypeShift) & 0xff;
                    if (todo) {
                        nx = x; ny = y; nh = h; nv = v; nvx = vx; nvy = vy;
                        moved = false; has_vel = false;
                    }
                    if constexpr (IDMF) {   // IDMController.step for the controlled lanes, on the state this step starts from
                        if (__ballot(todo) != 0ull) {
                            const bool act_now = i_valid && ((ids >> kIdsActiveShift) & 0xffu);
                            const double qnan = __builtin_nan("");
                            s_ixy[t] = act_now ? make_double2((double)x, (double)y) : make_double2(qnan, qnan);
                            s_iv[t] = v;
                            wave_sync();   // (an env's slots are all this wave's)
                            int lead = -1;
                            if (todo && act_now && has_ctrl) {
                                double sn, cs;
                                sincos_det((double)h, sn, cs);
                                const int ibase = i_env_local << log2A;
                                lead = idm::find_leader<false>([&](int j) { return s_ixy[ibase + j]; }, a_A, crow, (double)x, (double)y, sn, cs);
                                double dx = 0.0, dy = 0.0, vl = 0.0;
                                if (lead >= 0) {
                                    dx = s_ixy[ibase + lead].x - (double)x;
                                    dy = s_ixy[ibase + lead].y - (double)y;
                                    vl = (double)s_iv[ibase + lead];
                                }
                                a0 = (float)idm::idm_law(crow, (double)v, lead >= 0, dx, dy, vl);
                                a1 = 0.0f;
                                as_global(ia->idm_act0_own)[i_idx] = a0;
                                as_global(ia->idm_act1_own)[i_idx] = a1;
                            }
                            if (todo && i_valid) as_global(ia->idm_leader)[i_idx] = lead;
                            wave_sync();   // (the table is rewritten by the next round / step)
                        }
                    }
                    if (todo && i_valid && ((ids >> kIdsActiveShift) & 0xffu) && model < T2D_MODEL_DRIFT) {
                        auto P = [&](int col) -> double { return s_partab[col * T2D_MAX_TYPES + type]; };
                        const bool pm = model == T2D_MODEL_POINTMASS;
                        const integ::StepOut o = integ::step_participant<(FUSE > 0 ? 1 : 0)>(
                            model, P, (double)x, (double)y, (double)h, (double)v, pm ? (double)vx : 0.0, pm ? (double)vy : 0.0,
                            (double)a0, (double)a1, interval_ms, ia->interval_s);
                        nx = (float)o.x; ny = (float)o.y; nh = (float)o.heading; nv = (float)o.speed;
                        moved = true;
                        has_vel = o.has_velocity;
                        if (o.has_velocity) {
                            nvx = (float)o.vx;
                            nvy = (float)o.vy;
                        }
                        app0 = (float)o.app0;
                        app1 = (float)o.app1;
                    }
                    if (k < n_steps) {   // the hand-over, written while the verdict is still out (again after a reset)
                        s_hand[k & 1][0][t] = nx;
                        s_hand[k & 1][1][t] = ny;
                        s_hand[k & 1][2][t] = nh;
                        s_hand_ids[k & 1][t] = i_valid ? ids : 0u;
                    }
                    if (decided) {
                        T2D_IMARK(2);   // (the second round: integrating again after a reset; the first lands on mark 0)
                        break;
                    }
                    T2D_IMARK(0);
                    pipe_wait(&s_seq_e[w], (uint32_t)k);
                    T2D_IMARK(1);
                    decided = true;
                    const bool done = i_valid && s_dec[i_env_local] != 0u;
                    if (__ballot(done) == 0ull) break;
                    if (done) {   // the env's episode ended with step k - 1: back to the snapshot (t2d_reset's copy)
                        const float r0 = as_global(ia->snap[0])[i_idx], r1 = as_global(ia->snap[1])[i_idx], r2 = as_global(ia->snap[2])[i_idx];
                        const float r3 = as_global(ia->snap[3])[i_idx], r4 = as_global(ia->snap[4])[i_idx], r5 = as_global(ia->snap[5])[i_idx];
                        const uint32_t rid = as_global(ia->snap_ids)[i_idx];
                        as_global(ia->x)[i_idx] = r0;
                        as_global(ia->y)[i_idx] = r1;
                        as_global(ia->heading)[i_idx] = r2;
                        as_global(ia->speed)[i_idx] = r3;
                        as_global(ia->vx)[i_idx] = r4;
                        as_global(ia->vy)[i_idx] = r5;
                        as_global(ia->ids)[i_idx] = rid;
                        x = r0; y = r1; h = r2; v = r3; vx = r4; vy = r5; ids = rid;
                    }
                    todo = done && k < n_steps;
                    if (k == n_steps) break;
                }
                if (k == n_steps) break;
                // commit step k: the progress word first -- the event wave reads the hand-over, not memory -- then the state arrays
                pipe_post(&s_seq_i[w], (uint32_t)k + 1u);
                if (moved) {
                    as_global(ia->x)[i_idx] = nx;
                    as_global(ia->y)[i_idx] = ny;
                    as_global(ia->heading)[i_idx] = nh;
                    as_global(ia->speed)[i_idx] = nv;
                    if (has_vel && (((ids >> kIdsModelShift) & 0xff) == T2D_MODEL_POINTMASS || (ia->out_mask & T2D_OUT_VELOCITY))) {
                        as_global(ia->vx)[i_idx] = nvx;
                        as_global(ia->vy)[i_idx] = nvy;
                    }
                    if (ia->out_mask & T2D_OUT_APPLIED) {
                        as_global(ia->applied0)[i_idx] = app0;
                        as_global(ia->applied1)[i_idx] = app1;
                    }
                }
                x = nx; y = ny; h = nh; v = nv;
                if (has_vel) {
                    vx = nvx;
                    vy = nvy;
                }
                T2D_IMARK(3);
                a0 = na0;
                a1 = na1;
            }
            return;
        }
    }
    const int tid_outer = tid;
    // LOOP: what a trip leaves behind for the next one stays in registers -- the state it stored (or restored), the env's
    // counters, and the next trip's actions, requested a trip ahead: a lone wave hides no load, and each of these was a
    // round trip to the L2 at the head of its dependent chain
    [[maybe_unused]] uint32_t c_ids = 0;
    [[maybe_unused]] float c_x = 0, c_y = 0, c_h = 0, c_v = 0, c_vx = 0, c_vy = 0, n_a0 = 0, n_a1 = 0;
    [[maybe_unused]] int c_cnt = 0, c_frame = 0;
    for (;;) {   // (one trip unless LOOP)
    // LOOP: the lane's coordinates are derived again on every trip from a laundered thread id -- as loop invariants every
    // lane mask built from them (agent == 0, agent < n_off, valid && ..., one per use) is hoisted and held in a scalar
    // register pair across the whole body (170 spilled scalars)
    if constexpr (loop_rot) {
        switch ((step_k + ((int)blockIdx.x >> 8)) & 3) {   // (s_setprio takes an immediate)
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
    } else if constexpr (LOOP && !PIPE && T2D_LOOP_PRIO == 2) {
        if (behind_first) __builtin_amdgcn_s_setprio(3);
    }
    int tid_l = tid_outer;
    if constexpr (LOOP) asm volatile("" : "+v"(tid_l));
    const int tid = tid_l;
    const int lane = tid & 63;
    const int env_local = SPLIT ? unit % EPB : tid >> log2A;
    const int slot0 = SPLIT ? 0 : env_local << log2A;   // first LDS slot of the lane's env
    const int agent = tid & (A_pad - 1);
    const int env = SPLIT ? unit : wg * EPB + env_local;
    const bool valid = env < a_n_env && agent < a_A;
    const int idx = valid ? env * a_A + agent : 0;
    uint32_t* const queue = s_queue[SPLIT ? role : (tid >> 6) + (role_b ? kWaves : 0)];
    int* const qcount = &s_qcount[SPLIT ? role : (tid >> 6) + (role_b ? kWaves : 0)];
    if constexpr (LOOP) {
        pvp = late_args();
        cfgp = (const __attribute__((address_space(4))) t2d_status_config*)((const __attribute__((address_space(4))) char*)late_args() + kCfgArgOffset);
    }
    const auto& gl = pv.geo_layout;
    // ---------------- phase 0: issue every global load, clear LDS tables ------------------
    uint32_t ids = 0;
    float fx = 0, fy = 0, fh = 0;
    float fv = 0, fa0 = 0, fa1 = 0;  // fused: speed and actions
    [[maybe_unused]] int i_ctrl = T2D_IDM_NONE;   // IDMF: the participant's controller
    float bxmin = 0, bxmax = 0, bymin = 0, bymax = 0;
    bool has_boundary = false;
    // the status epilogue's inputs, fetched now by the lane that will run it (agent 0): their latency
    // would otherwise sit, unhidden, at the very end of the wave
    int pre_cnt = 0, pre_frame = 0;
    const bool carried = LOOP && step_k > step_first;   // (LOOP: the second and later trips take their inputs from registers)
    // The plain chained form: what does not depend on the step before -- the type table, the workgroup's geometry record, this
    // step's actions -- is requested BEFORE the wait for the hand-off, so that the poll's round trip to the L2 and theirs overlap
    // (scripts/chain_timing.py: hand-off wait 0.8 us + start-up 2.1 us of a slot's 16.8 us were two memory round trips in a row)
    constexpr bool PREF = CHAIN && !LOOP && !SPLIT && !IDMF && FUSE >= 0 && T2D_CHAIN_PREFETCH;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: no assignment across address spaces)
    const int n_vec = a_geo ? a_stride >> 2 : 0;
    const T2D_GLOBAL u32x4* gsrc = (const T2D_GLOBAL u32x4*)(a_geo + (size_t)wg * a_stride);
    [[maybe_unused]] double pf_t0 = 0.0, pf_t1 = 0.0, pf_t2 = 0.0;
    [[maybe_unused]] float pf_a0 = 0.0f, pf_a1 = 0.0f;
    [[maybe_unused]] u32x4 pf_geo = {0u, 0u, 0u, 0u};
    [[maybe_unused]] const bool pref = PREF && nthreads == kBlock && n_vec <= nthreads;
    if constexpr (PREF) {
        if (pref) {
            constexpr int kTabP = kTabCols * T2D_MAX_TYPES;
            pf_t0 = a_params[ptid];
            pf_t1 = a_params[ptid + kBlock];
            if (ptid + 2 * kBlock < kTabP) pf_t2 = a_params[ptid + 2 * kBlock];
            if (tid < n_vec) pf_geo = gsrc[tid];
            if (valid) {
                const size_t ai = (size_t)idx * a_act_stride + (size_t)step_k * (size_t)pv.chain_act_step;
                pf_a0 = a_act0[ai];
                pf_a1 = a_act1[ai];
            }
        }
    }
    if (CHAIN && step_k > 0 && !carried) chain_wait();
    if constexpr (CHAIN) { T2D_MARK(18); }
    if (valid && !PIPE && (!SPLIT || role == 0)) {   // (SPLIT: wave 0 loads and integrates; the others get the new state through LDS)
        if (carried) {
            ids = c_ids; fx = c_x; fy = c_y; fh = c_h; fv = c_v; fa0 = n_a0; fa1 = n_a1;
        } else {
            ids = ld_state<MULTI>(a_ids + idx);
            fx = ld_state<MULTI>(a_x + idx);
            fy = ld_state<MULTI>(a_y + idx);
            fh = ld_state<MULTI>(a_h + idx);
        }
        if (FUSE >= 0) {
            const size_t ai = (size_t)idx * a_act_stride + (MULTI ? (size_t)step_k * (size_t)pv.chain_act_step : 0);
            if (!carried) {
                fv = ld_state<MULTI>(a_v + idx);
                if (PREF && pref) {
                    fa0 = pf_a0;
                    fa1 = pf_a1;
                } else {
                    fa0 = a_act0[ai];
                    fa1 = a_act1[ai];
                }
            }
            if (LOOP && step_k + 1 < step_end) {   // the next trip's actions: their latency overlaps this trip
                n_a0 = a_act0[ai + (size_t)pv.chain_act_step];
                n_a1 = a_act1[ai + (size_t)pv.chain_act_step];
            }
            if (!(IDMF && !PIPE) && a_idm && a_idm[idx] != T2D_IDM_NONE) {  // IDM lane while caller actions are bound
                fa0 = pv.own_act0[idx];
                fa1 = pv.own_act1[idx];
            }
            if constexpr (IDMF && !PIPE) i_ctrl = (int)as_global(pv.idm_ctrl_all)[idx];   // (its row is fetched behind the barrier)
        }
        if (FUSE < 0 && pv.boundary) {  // fused: fetched after the integrator (register pressure)
            const float4 b = reinterpret_cast<const float4*>(pv.boundary)[env];
            bxmin = b.x; bxmax = b.y; bymin = b.z; bymax = b.w;
            has_boundary = pv.boundary_valid ? pv.boundary_valid[env] != 0 : true;
        }
    }
    constexpr bool stage_tables = !LOOP;   // (a LOOP launch staged them ahead of its loop: they stay in LDS across the steps)
    if (FUSE >= 0 && stage_tables) {  // full parameter table -> LDS, loads issued together (one exposed latency)
        constexpr int kTab = kTabCols * T2D_MAX_TYPES;
        if (nthreads == kBlock) {   // the usual launch shape: 3 x 8 B per thread, no per-load bounds logic
            static_assert(kTab <= 3 * kBlock && kTab > 2 * kBlock, "staging below assumes 2 full rounds + a (possibly full) third one");
            double t0, t1, t2;
            if (PREF && pref) {
                t0 = pf_t0; t1 = pf_t1; t2 = pf_t2;
            } else {
                t0 = a_params[ptid]; t1 = a_params[ptid + kBlock];
                t2 = ptid + 2 * kBlock < kTab ? a_params[ptid + 2 * kBlock] : 0.0;
            }
            s_partab[ptid] = t0;
            s_partab[ptid + kBlock] = t1;
            if (ptid + 2 * kBlock < kTab) s_partab[ptid + 2 * kBlock] = t2;
        } else {                    // workgroups narrowed by the geometry budget (64 or 128 threads)
            double tstage[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                tstage[k] = 0.0;
                if (k * 64 < kTab && tid + k * nthreads < kTab) tstage[k] = pv.params[tid + k * nthreads];
            }
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k * 64 < kTab && tid + k * nthreads < kTab) s_partab[tid + k * nthreads] = tstage[k];
        }
    }
    double par_stage[2] = {0.0, 0.0};  // events only: 4 shape columns x 32 types, <= 2 per thread
    if (FUSE < 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + k * nthreads;
            if (q < 4 * T2D_MAX_TYPES && (k == 0 || nthreads < 4 * T2D_MAX_TYPES))
                par_stage[k] = pv.params[(T2D_P_SHAPE + q / T2D_MAX_TYPES) * T2D_MAX_TYPES + q % T2D_MAX_TYPES];
        }
    }
    // geometry record -> LDS, 16-B loads
    {
        // rounds of 16-B loads this record needs (wave-uniform): 1 for the metric scenes (<= 4 KiB per workgroup) --
        // the unrolled generic form spends more on its per-load bounds logic than on the loads
        const int rounds = stage_tables ? (n_vec + nthreads - 1) / nthreads : 0;
        u32x4 geo_stage0 = {0u, 0u, 0u, 0u};
        const int gtid = SPLIT ? ptid : tid;   // (the thread's own number: SPLIT's `tid` is the participant slot)
        if (rounds <= 1) {
            if (gtid < n_vec && rounds == 1) geo_stage0 = (PREF && pref) ? pf_geo : gsrc[gtid];
        } else {
            // big records (many envs or many polygons per workgroup, e.g. 32 parking lots = 30 KiB for 32 threads):
            // global_load_lds -- 16 B per lane straight into LDS at M0 + lane * 16, no staging registers -- so ALL
            // rounds are in flight together and one memory latency is exposed instead of one per 8 loads
            for (int k = 0; k < rounds; ++k) {
                const int q = gtid + k * nthreads;
                if (q < n_vec)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(gsrc + q),
                        (__attribute__((address_space(3))) void*)(s_geo + 4 * (k * nthreads + (gtid & ~63))), 16, 0, 0);
            }
        }
        if (use_hash_grid)
            for (int k = tid; k < EPB * H; k += nthreads) s_head[k] = -1;
        if (LOOP) {   // (every wave clears what is its envs' own: the waves of a LOOP launch do not wait for each other)
            if (agent == 0 && !role_b) s_env_or[env_local] = 0;
        } else if (gtid < kBlock / 2) {
            s_env_or[gtid] = 0;
        }
        // (PIPE = 2: two waves OR into a participant's word; its owner clears it at the END of a trip, before the step's
        // verdict goes out -- the lane wave cannot be in the next step's stage before that -- and here on the first trip only)
        if (!(PIPE == 2 && (role_b || carried))) s_flags[tid] = 0;
        if (SPLIT && role == 0) s_ids_new[tid] = ids;   // (0 for a slot without a participant)
        if (FUSE < 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = tid + k * nthreads;
                if (q < 4 * T2D_MAX_TYPES && (k == 0 || nthreads < 4 * T2D_MAX_TYPES)) s_partab[q] = par_stage[k];
            }
        }
        if (rounds <= 1) {
            if (gtid < n_vec && rounds == 1) reinterpret_cast<u32x4*>(s_geo)[gtid] = geo_stage0;
        } else {
            __builtin_amdgcn_s_waitcnt(0);  // the LDS-direct loads are tracked by vmcnt: all landed before the barrier
        }
    }
    // (a) tables cleared, type columns + geometry record staged; the later trips of a LOOP launch staged nothing and cleared
    // only what belongs to their own wave
    if (carried && log2A <= 6) wave_sync(); else __syncthreads();
    T2D_MARK(0);
    if constexpr (CHAIN) {
        // The fragment's checkpoint: the state its first step starts from, stored from the registers that hold it anyway
        // (every slot of the pool, active or not).  Not behind a fragment that failed -- the error word is final by then,
        // launches of one stream do not overlap -- so the failed fragment's own checkpoint survives whatever was enqueued
        // behind it, and that is what the host restores.  A failure of THIS fragment must not stop its own checkpoint: on a
        // grid larger than the device holds, a consumer of step 1 can post its failure while the last workgroups of step 0
        // have not stored theirs yet (ckpt_open: the recorded failure, if any, carries this fragment's tag).
        if (step_k == 0 && valid && (!SPLIT || role == 0)) {
            const KernargView ck = late_args();
            if (ckpt_open(ck)) {
                auto c_base = as_global(ck->ckpt);
                const size_t cn = (size_t)ck->N;
                c_base[idx] = __float_as_uint(fx);
                c_base[cn + idx] = __float_as_uint(fy);
                c_base[2 * cn + idx] = __float_as_uint(fh);
                c_base[3 * cn + idx] = __float_as_uint(fv);
                c_base[6 * cn + idx] = ids;
            }
        }
    }

    if (SPLIT) ids = s_ids_new[tid];
    if constexpr (PIPE) {   // this step's state, committed by the pair's integrator wave
        pipe_wait(&s_seq_i[tid >> 6], (uint32_t)step_k + 1u);
        ids = s_hand_ids[step_k & 1][tid];
        fx = s_hand[step_k & 1][0][tid];
        fy = s_hand[step_k & 1][1][tid];
        fh = s_hand[step_k & 1][2][tid];
    }
    bool active = valid && ((ids >> kIdsActiveShift) & 0xffu);
    const int type = (ids >> kIdsTypeShift) & 0xff;
    if constexpr (IDMF && !PIPE) {
        // IDMController.step ahead of the integrator (what t2d_step does with an idm_kernel launch in front of this one): the
        // env's positions and speeds go through LDS -- the vertex planes and the centre table, dead until the pose phase --
        // every controlled lane sweeps its env for the leader and evaluates the law (t2d_idm_dev.h)
        const double qnan = __builtin_nan("");
        s_v[0][tid] = active ? (double)fx : qnan;
        s_v[1][tid] = active ? (double)fy : qnan;
        s_cxy[0][tid] = fv;
        const bool has_ctrl = active && i_ctrl != T2D_IDM_NONE && i_ctrl < pv.idm_n_ctrl;
        if (log2A <= 6) wave_sync(); else __syncthreads();
        int lead = -1;
        if (has_ctrl) {
            const idm::IdmRow crow = idm::load_row(as_global(pv.idm_rows) + (size_t)i_ctrl * T2D_IDM_COLS);
            double sn, cs;
            sincos_det((double)fh, sn, cs);
            lead = idm::find_leader<false>([&](int j) { return make_double2(s_v[0][slot0 + j], s_v[1][slot0 + j]); }, a_A, crow,
                                           (double)fx, (double)fy, sn, cs);
            double dx = 0.0, dy = 0.0, vl = 0.0;
            if (lead >= 0) {
                dx = s_v[0][slot0 + lead] - (double)fx;
                dy = s_v[1][slot0 + lead] - (double)fy;
                vl = (double)s_cxy[0][slot0 + lead];
            }
            fa0 = (float)idm::idm_law(crow, (double)fv, lead >= 0, dx, dy, vl);
            fa1 = 0.0f;
            as_global(pv.idm_act0_own)[idx] = fa0;
            as_global(pv.idm_act1_own)[idx] = fa1;
        }
        if (valid) as_global(pv.idm_leader)[idx] = lead;
        if (log2A <= 6) wave_sync(); else __syncthreads();   // (the planes are the pose phase's from here on)
    }
    // (SingleTrackDrift and euler point-mass lanes were integrated by drift_kernel, launched before this one)
    if (!PIPE && FUSE >= 0 && active && ((ids >> kIdsModelShift) & 0xff) < T2D_MODEL_DRIFT && (!SPLIT || role == 0) && !(T2D_PROBE_SKIP & 64)) {
        // ---------------- fused physics: one PhysicsModelBase.step in registers ----------------
        const int model = (ids >> kIdsModelShift) & 0xff;
        auto P = [&](int col) -> double { return s_partab[col * T2D_MAX_TYPES + type]; };
        double pvx = 0.0, pvy = 0.0;
        if (model == T2D_MODEL_POINTMASS) {
            if (carried) {
                pvx = (double)c_vx;
                pvy = (double)c_vy;
            } else {
                const float lvx = ld_state<MULTI>(as_global(pv.vx) + idx), lvy = ld_state<MULTI>(as_global(pv.vy) + idx);
                pvx = (double)lvx;
                pvy = (double)lvy;
                // (a point mass's velocity is state: part of the fragment's checkpoint.  The step number through an empty asm:
                // the compiler otherwise keeps the lane mask of the test at the top alive down to here, in spilled scalars)
                int k0 = CHAIN ? (int)blockIdx.y : 1;
                if constexpr (CHAIN) asm volatile("" : "+s"(k0));
                if (CHAIN && k0 == 0) {
                    const KernargView ck = late_args();
                    if (ckpt_open(ck)) {
                        as_global(ck->ckpt)[4 * (size_t)ck->N + idx] = __float_as_uint(lvx);
                        as_global(ck->ckpt)[5 * (size_t)ck->N + idx] = __float_as_uint(lvy);
                    }
                }
            }
        }
        const integ::StepOut o = integ::step_participant<(FUSE > 0 ? 1 : 0), (!LOOP || CHAIN || T2D_LOOP_RESUM)>(
            model, P, (double)fx, (double)fy, (double)fh, (double)fv, pvx, pvy, (double)fa0, (double)fa1, interval_ms, pv.interval_s);
        fx = (float)o.x;
        fy = (float)o.y;
        fh = (float)o.heading;
        fv = (float)o.speed;
        st_out<!MULTI>(pv.x + idx, fx);
        st_out<!MULTI>(pv.y + idx, fy);
        st_out<!MULTI>(pv.heading + idx, fh);
        st_out<!MULTI>(pv.speed + idx, fv);
        if (LOOP && o.has_velocity) {
            c_vx = (float)o.vx;
            c_vy = (float)o.vy;
        }
        if (o.has_velocity && (model == T2D_MODEL_POINTMASS || (pv.out_mask & T2D_OUT_VELOCITY))) {
            st_out<!MULTI>(pv.vx + idx, (float)o.vx);
            st_out<!MULTI>(pv.vy + idx, (float)o.vy);
        }
        if (pv.out_mask & T2D_OUT_APPLIED) {
            st_out<!MULTI>(pv.applied0 + idx, (float)o.app0);
            st_out<!MULTI>(pv.applied1 + idx, (float)o.app1);
        }
    }
    if (SPLIT) {   // the new state, from wave 0 to the three waves that take the other event stages
        if (role == 0) {
            s_new[0][tid] = fx;
            s_new[1][tid] = fy;
            s_new[2][tid] = fh;
        }
        __syncthreads();
        fx = s_new[0][tid];
        fy = s_new[1][tid];
        fh = s_new[2][tid];
    }
    T2D_MARK(13);
    double pre_tp = 0.0;
    if (FUSE >= 0 && valid && pv.boundary) {  // L2-resident by now (16 B per env)
        const float4 b = reinterpret_cast<const float4*>(pv.boundary)[env];
        bxmin = b.x; bxmax = b.y; bymin = b.z; bymax = b.w;
        has_boundary = pv.boundary_valid ? pv.boundary_valid[env] != 0 : true;
    }
    // ---------------- phase 1: pose, out-of-bound, conservative fp32 box ------------------------
    // (build-defined, as in the oracle's t2do_collide: a participant whose pose is not finite -- a NaN action that went through
    // np.clip, an overflow -- takes no part in event detection: no flag of its own, and nobody collides with it)
    // (`active` itself from here on -- nothing below asks whether the slot holds a participant, only whether it takes part:
    // a second lane mask kept alive through the event phases cost the looping forms 2 % on the highway pool)
    active = active && __builtin_isfinite(fx) && __builtin_isfinite(fy) && __builtin_isfinite(fh);
    int kind = -1;
    float R32 = -1.0f;                                  // bounding radius + 5 mm; < 0 = inactive
    float box_lo_x = 0, box_hi_x = 0, box_lo_y = 0, box_hi_y = 0;  // encloses the pose (outward rounded)
    uint32_t f_own = 0;                                 // flags this lane decides alone
    bool lane_safe = false;                             // pose certified inside the union of the env's lanes
    int gcx = 0, gcy = 0;
    if (active) {
        const double cx = (double)fx, cy = (double)fy;
        // columns SHAPE, LENGTH, WIDTH, RESERVED0 (bounding radius) are consecutive in the table
        constexpr int c0 = FUSE >= 0 ? T2D_P_SHAPE : 0;
        kind = (int)s_partab[(c0 + 0) * T2D_MAX_TYPES + type];
        const double L = s_partab[(c0 + 1) * T2D_MAX_TYPES + type];
        const double W = s_partab[(c0 + 2) * T2D_MAX_TYPES + type];
        const double R = s_partab[(c0 + 3) * T2D_MAX_TYPES + type];  // bounding radius (host-computed)
        const double rad = 0.5 * W;
        R32 = (float)R + 5e-3f;
        double lo_x, hi_x, lo_y, hi_y;
        bool out = false;
        if (kind == T2D_SHAPE_OBB) {
            double s, c;
            sincos_det((double)fh, s, c);
            const double hl = 0.5 * L, hw = 0.5 * W;
            // The oracle's vertex k is  c * lx[k] - s * ly[k] + cx,  s * lx[k] + c * ly[k] + cy  with (lx, ly) = (hl, -hw),
            // (hl, hw), (-hl, hw), (-hl, -hw).  Rounding is symmetric in sign, so its eight products are +-(c hl), +-(s hw),
            // +-(s hl), +-(c hw) and its eight first sums +-u, +-w, +-p, +-q below: the same sixteen results from half the
            // operations.  The pose box likewise: fl(cx + t) is monotonic in t, so the smallest of the four x is
            // fl(cx - max(|u|, |w|)), bit for bit the minimum the comparisons found.
            const double chl = c * hl, shw = s * hw, shl = s * hl, chw = c * hw;
            const double u = chl + shw, w = chl - shw;      // vertex 0 / 1 minus the centre, x
            const double pp = shl - chw, qq = shl + chw;    // ... y
            const double ax[4] = {u + cx, w + cx, cx - u, cx - w};
            const double ay[4] = {pp + cy, qq + cy, cy - pp, cy - qq};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_v[2 * k][tid] = ax[k];
                s_v[2 * k + 1][tid] = ay[k];
            }
            const double mx = __builtin_fmax(__builtin_fabs(u), __builtin_fabs(w));
            const double my = __builtin_fmax(__builtin_fabs(pp), __builtin_fabs(qq));
            lo_x = cx - mx; hi_x = cx + mx;
            lo_y = cy - my; hi_y = cy + my;
            // OutBound.update: not boundary.contains(pose); touching from inside is contained
            if (has_boundary)
                out = lo_x < (double)bxmin || hi_x > (double)bxmax || lo_y < (double)bymin || hi_y > (double)bymax;
        } else {
            lo_x = cx - rad; hi_x = cx + rad; lo_y = cy - rad; hi_y = cy + rad;
            if (has_boundary)
                out = cx - rad < (double)bxmin || cx + rad > (double)bxmax || cy - rad < (double)bymin ||
                      cy + rad > (double)bymax;
        }
        if (out) f_own |= T2D_FLAG_OUT_BOUND;
        s_cxy[0][tid] = fx;
        s_cxy[1][tid] = fy;
        // fp32 box rounded outwards (relative 2^-23 + 1e-6 m): conservative for the AABB passes
        box_lo_x = (float)lo_x; box_lo_x -= __builtin_fabsf(box_lo_x) * 1.2e-7f + 1e-6f;
        box_hi_x = (float)hi_x; box_hi_x += __builtin_fabsf(box_hi_x) * 1.2e-7f + 1e-6f;
        box_lo_y = (float)lo_y; box_lo_y -= __builtin_fabsf(box_lo_y) * 1.2e-7f + 1e-6f;
        box_hi_y = (float)hi_y; box_hi_y += __builtin_fabsf(box_hi_y) * 1.2e-7f + 1e-6f;
#ifndef T2D_NO_SAFE_RECTS
        // Off-lane short cut: a pose whose (outward-rounded) box lies in one of the env's safe rectangles -- rectangles the
        // host found inside the union of the env's lanes, t2d_api.hip build_safe_rects -- is contained in the union: the
        // lane stages below never see it.  A certificate, not an approximation: poses it does not cover take the exact path.
        if (gl.has[1]) {
            const float4* sr = reinterpret_cast<const float4*>(s_geo + gl.off_safe) + env_local * kSafeRects;
            // box inside rectangle: max(xmin - lo_x, hi_x - xmax, ymin - lo_y, hi_y - ymax) <= 0, two packed additions
            // with the box's (-lo, hi) pairs, a max3 and a max per rectangle (the form of the box sweeps below)
            typedef float f2s __attribute__((ext_vector_type(2)));
            const f2s cxp = {-box_lo_x, box_hi_x}, cyp = {-box_lo_y, box_hi_y};
#pragma unroll
            for (int k = 0; k < kSafeRects; ++k) {
                const float4 r = sr[k];
                const f2s tx = {r.x, -r.y}, ty = {r.z, -r.w};
                const f2s dx = tx + cxp, dy = ty + cyp;
                const float m = __builtin_fmaxf(__builtin_fmaxf(dx.x, dx.y), __builtin_fmaxf(dy.x, dy.y));
                lane_safe |= m <= 0.0f;
            }
        }
#endif
        if (use_hash_grid) {
            gcx = (int)__builtin_floor(cx * pv.inv_cell);
            gcy = (int)__builtin_floor(cy * pv.inv_cell);
            const int b = env_local * H + (int)(cell_hash(gcx, gcy) & (uint32_t)(H - 1));
            s_next[tid] = atomicExch(&s_head[b], tid);
        }
    }
    s_kind[tid] = kind;
    s_type[tid] = (unsigned char)type;
    T2D_MARK(1);
    // an env that fits in a wave only ever reads its own wave's poses: no workgroup barrier needed
    if (log2A <= 6) wave_sync(); else __syncthreads();  // (b) poses (and grid lists) visible
    T2D_MARK(2);

    // radius of circle i / bounding radius of box i (the values the pose phase computed: 0.5 * width, the host's R)
    auto rad_of = [&](int i) -> double {
        constexpr int c0 = FUSE >= 0 ? T2D_P_SHAPE : 0;
        const int t = s_type[i];
        return s_kind[i] == T2D_SHAPE_OBB ? s_partab[(c0 + 3) * T2D_MAX_TYPES + t] : 0.5 * s_partab[(c0 + 2) * T2D_MAX_TYPES + t];
    };
    // ---------------- phase 2a: participant pairs ----------------------------------------------
    // narrow phase of one unordered pair (i < j by construction); flags both participants
    auto process_pair = [&](uint32_t e) {
        const int i = (int)(e & 255u), j = (int)(e >> 8);
        const int ki = s_kind[i], kj = s_kind[j];
        bool hit;
        if (ki == T2D_SHAPE_OBB && kj == T2D_SHAPE_OBB) {
            const Quad A = load_obb_lds(&s_v[0][i]), B = load_obb_lds(&s_v[0][j]);
#ifndef T2D_NO_RECT_FILTER
            // two boxes: four projections certify the answer unless the boxes are within ~1e-7 m of touching
            // (rect_pair_filter); only then -- practically never -- the wave runs the 32 orientations of the oracle's test
            const int v = rect_pair_filter(A, B);
            hit = v == 1;
            if (__ballot(v == 2) != 0ull) {
                if (v == 2) hit = sat_quads(A, B);
            }
#else
            hit = sat_quads(A, B);
#endif
        } else if (ki == T2D_SHAPE_OBB) {   // circle j against box i
            hit = circle_vs_generic((double)s_cxy[0][j], (double)s_cxy[1][j], rad_of(j), PolyRef{nullptr, &s_v[0][i], 4});
        } else if (kj == T2D_SHAPE_OBB) {   // circle i against box j
            hit = circle_vs_generic((double)s_cxy[0][i], (double)s_cxy[1][i], rad_of(i), PolyRef{nullptr, &s_v[0][j], 4});
        } else {                            // circle - circle (oracle: c1 = i, c2 = j)
            const double dx = (double)s_cxy[0][i] - (double)s_cxy[0][j], dy = (double)s_cxy[1][i] - (double)s_cxy[1][j];
            const double rr = rad_of(i) + rad_of(j);
            hit = dx * dx + dy * dy <= rr * rr;
        }
        if (hit) {
            atomicOr(&s_flags[i], T2D_FLAG_COLLISION_DYNAMIC);
            atomicOr(&s_flags[j], T2D_FLAG_COLLISION_DYNAMIC);
        }
    };
    const KernargView ev_args = late_args();   // the event stages' layout offsets are fetched here, not at the kernel's entry
    const auto& gl2 = ev_args->geo_layout;
    const int* geo_i = reinterpret_cast<const int*>(s_geo);
    auto process_static = [&](uint32_t e) {
        const int i = (int)(e & 255u), p = (int)(e >> 8);
        const int* vstart = geo_i + gl2.off_vstart[0];
        const float* xy = reinterpret_cast<const float*>(s_geo + gl2.off_xy[0]);
        const int v0 = vstart[p], n = vstart[p + 1] - v0;
        bool hit;
        if (s_kind[i] == T2D_SHAPE_OBB) {
            // (3 or 4 vertices: t2d_set_static_geometry cuts larger polygons into fans of quads)
#ifndef T2D_NO_RECT_FILTER
            // most candidates of the box sweep are clear of the polygon: an edge of the polygon with the whole box beyond it
            // certifies that, a box whose centre lies inside the polygon is a hit (rect_vs_convex_filter); the rest -- touching, overlapping, or separated only by an edge of the
            // box -- take the oracle's 32 orientations (both quads are read again there: held across the filter they cost
            // the kernel 19 spilled registers)
            const int v = rect_vs_convex_filter(load_obb_lds(&s_v[0][i]), load_quad_f32(xy + 2 * v0, n));
            hit = v == 1;
            if (__ballot(v == 2) != 0ull) {
                asm volatile("" ::: "memory");
                if (v == 2) hit = sat_quads(load_obb_lds(&s_v[0][i]), load_quad_f32(xy + 2 * v0, n));
            }
#else
            hit = sat_quads(load_obb_lds(&s_v[0][i]), load_quad_f32(xy + 2 * v0, n));
#endif
        } else {
            hit = circle_vs_generic((double)s_cxy[0][i], (double)s_cxy[1][i], rad_of(i),
                                    PolyRef{reinterpret_cast<const float2*>(xy + 2 * v0), nullptr, n});
        }
        if (hit) atomicOr(&s_flags[i], T2D_FLAG_COLLISION_STATIC);
    };
    // off-lane = not union(lanes).contains(pose), the oracle's definition (box_in_lane_union / circle_in_lane_union):
    //     contained  <=>  the centre lies in some lane polygon  and  no boundary piece of the union meets the open pose.
    // One pass per (participant, lane polygon whose box meets the pose's) candidate: is the centre in this polygon (bit 0 of
    // s_flags >> kLaneShift), does one of the union's boundary pieces that are part of this polygon's edges meet the open pose
    // (bit 1).  Every polygon holding the centre and every piece that can reach the pose belongs to such a candidate (a piece
    // lies inside its polygon's box), so the OR over the candidates is the oracle's verdict over all polygons and pieces.
    // (Rounds 1-2 ran two short cuts first -- all four vertices in one polygon, a vertex in none: 16 orientations per
    // candidate, then a second compacted pass for the bodies that straddle lanes.  Same verdicts: 0 of 147 k poses differ.)
    // A piece is first held against the pose's SUPPORT along the piece's normal n: the pose lies on one side of the line AB,
    // clear of it, when |n.(c - A)| exceeds (|n.P| + |n.Q|) / 2 (P, Q the box's edge vectors) -- then all four orientations
    // orient(A, B, vertex) have one sign and piece_meets_quad_interior returns false by its second rule; the margin (1e-9 |n|,
    // a nanometre) is a thousand times their rounding.  Only pieces that run through or touch the pose take the exact test.
    auto process_lane = [&](uint32_t e) {
        const int i = (int)(e & 255u), p = (int)(e >> 8);
        const int* vstart = geo_i + gl2.off_vstart[1];
        const int* bstart = geo_i + gl2.off_bstart;
        const float* xy = reinterpret_cast<const float*>(s_geo + gl2.off_xy[1]);
        const int v0 = vstart[p], n = vstart[p + 1] - v0;
        const int b0 = bstart[p], b1 = bstart[p + 1];
        const double2* bnd = reinterpret_cast<const double2*>(s_geo + gl2.off_bnd);
        const double cx = (double)s_cxy[0][i], cy = (double)s_cxy[1][i];
        uint32_t bits = point_in_quad(load_quad_f32(xy + 2 * v0, n), cx, cy) ? 1u : 0u;   // 3 or 4 vertices (fans of quads)
        bool cut = false;
        if (s_kind[i] == T2D_SHAPE_OBB) {
            const Quad A = load_obb_lds(&s_v[0][i]);
            const double px = A.x[0] - A.x[3], py = A.y[0] - A.y[3], qx = A.x[1] - A.x[0], qy = A.y[1] - A.y[0];
            const double c2x = A.x[0] + A.x[2], c2y = A.y[0] + A.y[2];   // twice the centre
            for (int b = b0; b < b1; ++b) {
                const double2 Pa = bnd[2 * b], Pb = bnd[2 * b + 1];
                const double nx = Pa.y - Pb.y, ny = Pb.x - Pa.x;
                const double s2 = __builtin_fma(nx, c2x - 2.0 * Pa.x, ny * (c2y - 2.0 * Pa.y));
                const double e2 = __builtin_fabs(__builtin_fma(nx, px, ny * py)) + __builtin_fabs(__builtin_fma(nx, qx, ny * qy));
                const bool clear = __builtin_fabs(s2) > e2 + 2e-9 * (__builtin_fabs(nx) + __builtin_fabs(ny));
                if (__ballot(!clear) != 0ull) {
                    if (!clear) cut |= piece_meets_quad_interior(A, Pa.x, Pa.y, Pb.x, Pb.y);
                }
            }
        } else {   // pedestrian: a piece inside the open disc.  The line AB further than R from the centre: so is the piece
            const double R = rad_of(i), R2 = R * R;
            for (int b = b0; b < b1; ++b) {
                const double2 Pa = bnd[2 * b], Pb = bnd[2 * b + 1];
                const double nx = Pa.y - Pb.y, ny = Pb.x - Pa.x;
                const double sn = __builtin_fma(nx, cx - Pa.x, ny * (cy - Pa.y));
                const bool clear = sn * sn > R2 * __builtin_fma(nx, nx, ny * ny) * (1.0 + 1e-9);
                if (__ballot(!clear) != 0ull) {
                    if (!clear) cut |= seg_dist2(Pa.x, Pa.y, Pb.x, Pb.y, cx, cy) < R2;
                }
            }
        }
        if (cut) bits |= 2u;
        if (bits) atomicOr(&s_flags[i], bits << kLaneShift);
    };
    int n_lane_polys = 0;
    // Some waves visit the polygon stages before the pair stage: the four waves of a SIMD start together and would
    // otherwise sit in the same latency-bound (LDS compaction) or issue-bound (narrow phase) stretch at the same time.
    // The stages are independent (results are OR-ed into s_flags).  (The workgroups of a launch go round the XCDs, then
    // round an XCD's 32 CUs: workgroups b, b + 256, b + 512, b + 768 share a CU, and a SIMD holds one wave of each.)
    // Which waves take the polygon stages first.  The launch that has the GPU to itself: the third and fourth of the four
    // workgroups that share a CU (a SIMD holds one wave of each) -- measured against none 25.3, all 24.6, every other
    // workgroup 24.5, alternating waves 25.1 us per step: 24.2.  Overlapping launches of env groups (256 workgroups
    // each): alternating waves and workgroup rounds, as before.
    const bool polys_first = log2A <= 6 && (pv.overlapped ? ((((int)blockIdx.x >> 8) + (tid >> 6)) & 1) != 0
                                                          : (((int)blockIdx.x >> 9) & 1) != 0);
    if (gl2.has[1] && !(T2D_PROBE_SKIP & 8)) {
        const int* pstart = geo_i + gl2.off_pstart[1];
        n_lane_polys = pstart[env_local + 1] - pstart[env_local];
    }
    // (SPLIT: one pass -- wave 0 takes the pair stage, the others the polygon stages, each its part: see below)
    for (int stage_it = 0; stage_it < (SPLIT ? 1 : 2); ++stage_it) {
    if (!pipe_prio && !(CHAIN && chain_late)) {
        if (behind_first && stage_it == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
    }
    if (SPLIT ? role == 0 : ((stage_it == 0) != polys_first && !role_b)) {
    if (T2D_PROBE_SKIP & 1) {
    } else if (!use_hash_grid) {
        // every lane against all agents of its env: fp32 circles, 1 cm margin (strictly
        // conservative for |x|,|y| < 4 km); executed by ALL lanes (shuffles read executing lanes)
        unsigned long long cand = 0ull;
        if (log2A > 0) {
            // (x, y, R) of the wave's 64 lanes go through LDS (the wave's queue storage, idle until the
            // compaction below) and come back two agents at a time as wave-uniform 8-B reads; the
            // distance test runs on packed fp32 (v_pk_*: 2 agents per instruction) and each verdict is
            // shifted into the lane's candidate word by v_cmp + v_addc (carry-in = the compare mask):
            // 5 VALU instructions per agent instead of 11.  Agents are visited in descending order so
            // that agent a lands on bit a.  Inactive lanes are parked at x = 1e30 (distance^2 = inf).
            typedef float f2 __attribute__((ext_vector_type(2)));
            float* const bx = reinterpret_cast<float*>(queue);
            float* const by = bx + 96;
            float* const bR = by + 96;
            static_assert(3 * 96 <= kQueueCap, "broad-phase staging must fit in the wave's queue");
            // (the centre comes back from LDS: carried in registers since the pose phase it cost the kernel a spill)
            const float fx_b = s_cxy[0][tid], fy = s_cxy[1][tid];
            const float px = active ? fx_b : 1e30f;
            // every env of the wave gets a segment of 1.5 * A_pad entries: its agents, then its first half again, so
            // that the ring sweep below reads agent + offset without wrapping (64 / A_pad segments: 96 entries in all)
            const int n_off = A_pad >> 1;                           // partners per lane: agent + 1 .. agent + A_pad / 2
            const int slot = (lane >> log2A) * (A_pad + n_off) + agent;
            bx[slot] = px;
            by[slot] = fy;
            bR[slot] = R32;
            if (agent < n_off) {
                bx[slot + A_pad] = px;
                by[slot + A_pad] = fy;
                bR[slot + A_pad] = R32;
            }
            wave_sync();
            const f2 px2 = {px, px}, py2 = {fy, fy}, pR2 = {R32, R32};
            // Ring sweep: lane (agent a) looks at the n_off partners a + 1 .. a + n_off round its env, so every unordered
            // pair is tested once (offset d lands on bit d - 1; offset n_off is seen from both ends and kept by the
            // lower half).  K pairs of offsets per batch: all 3K (unaligned 2 x 4 B) LDS reads are issued before the
            // first compare, so one LDS latency is exposed per batch instead of per pair.
            uint32_t h = 0u;
            auto batch = [&](uint32_t& hh, int top, auto kc) {   // offsets top + 2K - 1 ... top
                constexpr int K = decltype(kc)::value;
                f2 ox[K], oy[K], oR[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int i0 = slot + top + 2 * (K - 1 - k);   // offsets i0 - slot (.x) and i0 - slot + 1 (.y)
                    ox[k] = f2{bx[i0], bx[i0 + 1]};
                    oy[k] = f2{by[i0], by[i0 + 1]};
                    oR[k] = f2{bR[i0], bR[i0 + 1]};
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const f2 dx = px2 - ox[k], dy = py2 - oy[k], rr = pR2 + oR[k];
                    const f2 q = __builtin_elementwise_fma(dy, dy, dx * dx), r = rr * rr;
                    asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                 : "+v"(hh) : "v"(q.y), "v"(r.y) : "vcc");
                    asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                 : "+v"(hh) : "v"(q.x), "v"(r.x) : "vcc");
                }
            };
            using K1 = std::integral_constant<int, 1>;
            using K4 = std::integral_constant<int, 4>;
            if (log2A == 6) {          // env == wave: constant trip count, fully unrolled
#pragma unroll
                for (int top = 25; top >= 1; top -= 8) batch(h, top, K4{});
            } else if (n_off >= 8) {   // A_pad = 16 or 32
                for (int top = n_off - 7; top >= 1; top -= 8) batch(h, top, K4{});
            } else if (n_off >= 2) {   // A_pad = 4 or 8
                for (int top = n_off - 1; top >= 1; top -= 2) batch(h, top, K1{});
            } else {                   // A_pad = 2: the one partner
                const float dx = px - bx[slot + 1], dy = fy - by[slot + 1], rr = R32 + bR[slot + 1];
                h = (dx * dx + dy * dy <= rr * rr) ? 1u : 0u;
            }
            if (agent >= n_off) h &= ~(1u << (n_off - 1));   // the pair at offset n_off belongs to its lower agent
            cand = h;
        }
        T2D_MARK(3);
        if (!active || (T2D_PROBE_SKIP & 2)) cand = 0ull;
        compact_and_process<false, true>(cand, slot0, tid, queue, qcount, lane, process_pair, true, A_pad - 1);
    } else {
        // spatial-hash walk; pairs go straight to the narrow phase (i < j de-duplicates)
        if (active) {
            const double cx = (double)s_cxy[0][tid], cy = (double)s_cxy[1][tid];
            for (int oy_ = -1; oy_ <= 1; ++oy_)
                for (int ox_ = -1; ox_ <= 1; ++ox_) {
                    const int b = env_local * H + (int)(cell_hash(gcx + ox_, gcy + oy_) & (uint32_t)(H - 1));
                    for (int j = s_head[b]; j >= 0; j = s_next[j]) {
                        if (j <= tid) continue;
                        const int jx = (int)__builtin_floor((double)s_cxy[0][j] * pv.inv_cell);
                        const int jy = (int)__builtin_floor((double)s_cxy[1][j] * pv.inv_cell);
                        if (jx != gcx + ox_ || jy != gcy + oy_) continue;  // hash alias of another cell
                        const double dx = cx - (double)s_cxy[0][j], dy = cy - (double)s_cxy[1][j];
                        const double rr = rad_of(tid) + rad_of(j) + kRejectMargin;
                        if (dx * dx + dy * dy > rr * rr) continue;  // cannot touch
                        process_pair((uint32_t)tid | ((uint32_t)j << 8));
                    }
                }
        }
    }

    T2D_MARK(4);
    if (use_hash_grid) __syncthreads();  // the grid lists share LDS with the queues used below
    // ---------------- phase 2b / 2c: static polygons and lane polygons -------------------------
    } else if (!role_b || (stage_it == 0) == polys_first) {
    {
        // Broad -> narrow stages over boxes kept in the LDS record (static polygons, lane polygons, boundary pieces of the
        // lane union): pass 1 = which (participant, box) pairs meet; pass 2 = the survivors of the whole wave, compacted,
        // one narrow test per lane.
        // box-vs-box sweep of boxes [first, first + cn) of `bb`: the box's float4 (xmin, xmax, ymin, ymax) comes from the LDS
        // record (one 16-B read, wave-uniform address when the env is a wave), the four inequalities collapse into
        // max(xmin - hi_x, lo_x - xmax, ymin - hi_y, lo_y - ymax) <= 0 -- two packed subtractions with the participant's
        // (-hi, lo) pairs, one max3, one max -- and v_cmp + v_addc shifts the verdict into the mask (descending order, so
        // box q lands on bit q).  6 VALU per box instead of ~11.  (ox, oy) = (-hi_x, lo_x), (-hi_y, lo_y) of the
        // participant's box, or of a point.
        typedef float f2p __attribute__((ext_vector_type(2)));
        auto box_sweep = [&](const float4* bb, int first, int cn, const f2p bxp, const f2p byp) -> unsigned long long {
            uint32_t hw[2] = {0u, 0u};
#pragma unroll
            for (int hb = 1; hb >= 0; --hb) {
                uint32_t h = 0u;
                const int top = cn - hb * 32 < 32 ? cn - hb * 32 : 32;   // boxes of this half-word
                auto verdict = [&](const float4 b) {
                    const f2p tx = {b.x, -b.y}, ty = {b.z, -b.w};
                    const f2p dx = tx + bxp, dy = ty + byp;   // (xmin - hi_x, lo_x - xmax), (ymin - hi_y, lo_y - ymax)
                    const float m = __builtin_fmaxf(__builtin_fmaxf(dx.x, dx.y), __builtin_fmaxf(dy.x, dy.y));
                    asm volatile("v_cmp_ge_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(h) : "v"(m) : "vcc");
                };
                // the (top & 3) highest boxes one at a time, the rest four per trip with the four 16-B reads issued
                // before the first compare: one LDS round trip per four boxes instead of one each
                int a = top - 1;
                for (; (a & 3) != 3 && a >= 0; --a) verdict(bb[first + hb * 32 + a]);
                for (; a >= 3; a -= 4) {
                    const float4* q = bb + first + hb * 32 + a;
                    const float4 b0 = q[0], b1 = q[-1], b2 = q[-2], b3 = q[-3];
                    verdict(b0);
                    verdict(b1);
                    verdict(b2);
                    verdict(b3);
                }
                hw[hb] = h;
            }
            return (unsigned long long)hw[0] | ((unsigned long long)hw[1] << 32);
        };
        // boxes [lo, hi) of `bb` against the lanes that `want` the test; entries (participant | box index << 8) go to `process`
        auto sweep_stage = [&](const float4* bb, int lo, int hi, bool want, const f2p ox, const f2p oy, auto process, int mark) {
            if (log2A == 6) {
                const int np = hi - lo;  // wave-uniform
                for (int c0 = 0; c0 < np; c0 += 64) {
                    const int cn = np - c0 < 64 ? np - c0 : 64;
                    unsigned long long m = box_sweep(bb, lo + c0, cn, ox, oy);
                    if (!want) m = 0ull;
                    T2D_MARK(mark);
                    compact_and_process<false>(m, lo + c0, tid, queue, qcount, lane, process, cn <= 32);
                    T2D_MARK(mark + 1);
                }
            } else {
                for (int c0 = 0;; c0 += 64) {
                    const int left = want ? hi - lo - c0 : 0;
                    if (__ballot(left > 0) == 0ull) break;
                    const int cn = left < 64 ? left : 64;
                    const unsigned long long m = cn > 0 ? box_sweep(bb, lo + c0, cn, ox, oy) : 0ull;
                    T2D_MARK(mark);
                    compact_and_process<false>(m, lo + c0, tid, queue, qcount, lane, process);
                    T2D_MARK(mark + 1);
                }
            }
        };
        const f2p bxp = {-box_hi_x, box_lo_x}, byp = {-box_hi_y, box_lo_y};
        if (gl2.has[0] && !(T2D_PROBE_SKIP & 4) && (!SPLIT || role == 1) && !role_b) {   // static obstacles
            const int* pstart = geo_i + gl2.off_pstart[0];
            sweep_stage(reinterpret_cast<const float4*>(s_geo + gl2.off_aabb[0]), pstart[env_local], pstart[env_local + 1], active,
                        bxp, byp, process_static, 5);
        }
        if (gl2.has[1] && !(T2D_PROBE_SKIP & 8) && (!SPLIT || role >= 2) && (PIPE != 2 || role_b)) {   // lanes: off-lane = not union(lanes).contains(pose)
            const int* pstart = geo_i + gl2.off_pstart[1];
            int p0 = pstart[env_local], p1 = pstart[env_local + 1];
            if (SPLIT) {   // waves 2 and 3: the first and the second half of the env's lane polygons
                const int mid = p0 + ((p1 - p0 + 1) >> 1);
                if (role == 2) p1 = mid; else p0 = mid;
            }
            const bool want = active && !lane_safe && p1 > p0;   // (poses in a safe rectangle are done)
            if (__ballot(want) != 0ull)   // (wave-uniform: the lanes of a wave may belong to different envs)
                sweep_stage(reinterpret_cast<const float4*>(s_geo + gl2.off_aabb[1]), p0, p1, want, bxp, byp, process_lane, 7);
        }
    }

    }
    }
    if (behind_first && !(CHAIN && chain_late)) __builtin_amdgcn_s_setprio(0);
    // ---------------- phase 3: reduce + status epilogue ------------------------------------
    // every kernel argument the epilogue touches, requested together (see the start-up phase): the status lane's chain
    // and the restore of a finished env are the last thing a wave does, with nothing behind them to hide a scalar round
    // trip per pointer
    const KernargView ep = late_args();
    auto e_flags = as_global(ep->flags);
    auto e_env_flags = as_global(ep->env_flags);
    auto e_cnt_step = as_global(ep->cnt_step);
    auto e_frame_ms = as_global(ep->frame_ms);
    auto e_status = as_global(ep->status);
    auto e_reward = as_global(ep->reward);
    auto e_record = as_global(ep->record);
    auto e_time_penalty = as_global(ep->time_penalty);
    auto e_iou = as_global(ep->iou);
    auto e_last_valid = as_global(ep->last_valid);
    auto e_cnt_na = as_global(ep->cnt_na);
    auto e_max_iou = as_global(ep->max_iou);
    auto e_min_dist = as_global(ep->min_dist);
    auto e_snap_min_dist = as_global(ep->snap_min_dist);
    int e_auto_reset = ep->auto_reset;
    auto e_record_ring = as_global(ep->record_ring);
    auto e_target_c = as_global(ep->target_c);
    int e_record_slot0 = ep->record_slot0, e_n_env = ep->n_env;
    asm volatile("" : "+s"(e_flags), "+s"(e_env_flags), "+s"(e_cnt_step), "+s"(e_frame_ms), "+s"(e_status), "+s"(e_reward), "+s"(e_record), "+s"(e_time_penalty), "+s"(e_iou), "+s"(e_last_valid), "+s"(e_cnt_na), "+s"(e_max_iou), "+s"(e_min_dist), "+s"(e_snap_min_dist), "+s"(e_auto_reset), "+s"(e_record_ring), "+s"(e_target_c), "+s"(e_record_slot0), "+s"(e_n_env));
    // LOOP: the status configuration as well, all sixteen words in one scalar round trip.  Its fields are read through the
    // per-trip laundered pointer (see the top of the loop), so left alone every one is fetched where it is first used --
    // behind its branch of the status logic: eight to ten dependent scalar round trips in a row at the end of the wave's
    // chain (-DT2D_TIMING: 1.9 k cycles of epilogue per step on the highway pool, most of it these).
    t2d_status_config lcfg;
    if constexpr (!LOOP) {
        lcfg = cfg_arg;
    } else {   // (field by field: the source sits in the constant address space)
        lcfg.max_step = cfg.max_step; lcfg.ego_index = cfg.ego_index; lcfg.check_dynamic = cfg.check_dynamic;
        lcfg.check_off_lane = cfg.check_off_lane; lcfg.reward_collision = cfg.reward_collision;
        lcfg.reward_time_exceed = cfg.reward_time_exceed; lcfg.reward_out_bound = cfg.reward_out_bound;
        lcfg.reward_completed = cfg.reward_completed; lcfg.time_penalty_scale = cfg.time_penalty_scale;
        lcfg.check_arrival = cfg.check_arrival; lcfg.check_no_action = cfg.check_no_action;
        lcfg.no_action_max_step = cfg.no_action_max_step; lcfg.shaped_reward = cfg.shaped_reward;
        lcfg.arrival_threshold = cfg.arrival_threshold; lcfg.no_action_iou = cfg.no_action_iou;
        lcfg.dist_reward_scale = cfg.dist_reward_scale;
        asm volatile("" : "+s"(lcfg.max_step), "+s"(lcfg.ego_index), "+s"(lcfg.check_dynamic), "+s"(lcfg.check_off_lane),
                          "+s"(lcfg.reward_collision), "+s"(lcfg.reward_time_exceed), "+s"(lcfg.reward_out_bound),
                          "+s"(lcfg.reward_completed), "+s"(lcfg.time_penalty_scale), "+s"(lcfg.shaped_reward),
                          "+s"(lcfg.dist_reward_scale));
    }
#undef cfg
#define cfg lcfg
    // the status epilogue's inputs, requested now by the lane that will run it (agent 0): the reduce hides part of their
    // latency.  (Fetched at the top of the kernel they sat in registers through every event phase, and at the 128
    // registers of 4 waves / SIMD that meant scratch spills: 8 B per lane stored and re-read, 13 MB of HBM traffic.)
    if (WITH_STATUS && valid && agent == 0 && (!SPLIT || role == 0) && !role_b) {
        if (carried) {
            pre_cnt = c_cnt;
            pre_frame = c_frame;
        } else {
            pre_cnt = ld_state<MULTI>(e_cnt_step + env);
            pre_frame = ld_state<MULTI>(e_frame_ms + env);
            int k0 = CHAIN ? (int)blockIdx.y : 1;
            if constexpr (CHAIN) asm volatile("" : "+s"(k0));
            if (CHAIN && k0 == 0 && ckpt_open(ep)) {   // (the env's counters at the start of the fragment: the checkpoint's last two columns)
                auto c_env = as_global(ep->ckpt) + 7 * (size_t)ep->N;
                c_env[env] = (uint32_t)pre_cnt;
                c_env[e_n_env + env] = (uint32_t)pre_frame;
            }
        }
        if (e_time_penalty && cfg.max_step > 0) {
            const int c = pre_cnt + 1;
            pre_tp = e_time_penalty[c < cfg.max_step ? c : cfg.max_step];
        }
    }
    T2D_MARK(9);
    if (log2A <= 6 && !SPLIT) wave_sync(); else __syncthreads();  // (c) every queue drained: s_flags complete
    if constexpr (PIPE == 2) {   // ... the lane wave's too
        if (role_b) pipe_post(&s_seq_b[tid >> 6], (uint32_t)step_k + 1u);
        else pipe_wait(&s_seq_b[tid >> 6], (uint32_t)step_k + 1u);
    }
    T2D_MARK(10);
    if ((!SPLIT || role == 0) && !role_b) {   // (SPLIT: the reduce, the epilogue and the restore are wave 0's)
    uint32_t f = 0;
    if (active) {
        const uint32_t sf = s_flags[tid];
        f = f_own | (sf & ((1u << kLaneShift) - 1u));
        if constexpr (FUSE < 0) {   // maps in the HBM grid tier: the map events' verdicts for this participant (t2d_mapgrid.hip)
            if (pv.map_flags) f |= pv.map_flags[idx];
        }
        if (n_lane_polys > 0) {  // build-defined off-lane: not union(lanes).contains(pose)
            const uint32_t b = sf >> kLaneShift;   // see process_lane
            const bool in = lane_safe || ((b & 1u) && !(b & 2u));
            if (!in) f |= T2D_FLAG_OFF_LANE;
        }
    }
    if (valid) st_out<!MULTI>(e_flags + idx, f);
    s_flags[tid] = f;
    if (__ballot(f != 0) != 0ull && f != 0) atomicOr(&s_env_or[env_local], f);
    if (log2A <= 6) wave_sync(); else __syncthreads();  // (d)
    T2D_MARK(11);

    // IoU pre-pass: the two quad-IoUs an ego may need -- NoAction (pose vs previous pose) and Arrival (pose vs target
    // bay) -- are evaluated by lanes 0 and 1 of the env IN THE SAME SIMT PASS (one trip through quad_iou, not two in a
    // row on one lane: ~5 us of the ~22 us parking step).  Whether the values are used is decided by the epilogue
    // below exactly as before; an unused value is simply dropped.
    double iou_na = 0.0, iou_ar = 0.0;
    if (WITH_STATUS && IOU && (cfg.check_no_action || cfg.check_arrival)) {
        const int ego_l = slot0 + cfg.ego_index;
        const bool env_ok = env < pv.n_env && agent < 2;
        bool want = false;
        const double* other = nullptr;
        if (env_ok && s_kind[ego_l] == T2D_SHAPE_OBB) {
            if (agent == 0 && cfg.check_no_action && e_last_valid[env]) {
                want = true;
                other = pv.last_pose + 8 * (size_t)env;
            } else if (agent == 1 && cfg.check_arrival && pv.target_xy) {
                want = true;
                other = pv.target_xy + 8 * (size_t)env;
            }
        }
        double v = 0.0;
        if (__ballot(want) != 0ull) {
            if (want) v = quad_iou(&s_v[0][ego_l], other);
        }
        iou_na = v;
        iou_ar = __shfl_down(v, 1);   // lane 0 of the env receives lane 1's value (same wave: 2^log2A >= 2 lanes per env)
    }
    // One env per wave (64-participant envs) without IoU events, shaped reward or the tanh fall-back: everything the status
    // epilogue works on is ONE value per wave.  Made wave-uniform by v_readfirstlane, the ordered early-return logic of
    // check_status, the reward table and the counters compile to SCALAR instructions -- the scalar pipe has slack (4.2
    // cycles per instruction beside the VALU work of other waves, profiles/valu_issue_cycles.json) while the VALU is what
    // bounds the launch: the wave issued ~100 VALU instructions here for the sake of one lane, now ~20 (the stores' data).
    // Same decisions, same bits: the generic path below, instruction for instruction, on the same inputs.
    bool epilogue_done = false;
    if constexpr (WITH_STATUS && !IOU && !SPLIT) {
#ifndef T2D_NO_SCALAR_EPILOGUE
        if (log2A == 6 && !cfg.shaped_reward && (cfg.max_step <= 0 || e_time_penalty) && !(T2D_PROBE_SKIP & 128)) {
            epilogue_done = true;
            if (valid && agent == 0 && !role_b) {
                const int env_s = __builtin_amdgcn_readfirstlane(env);
                const uint32_t env_or = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_env_or[env_local]);
                const uint32_t ef = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flags[slot0 + cfg.ego_index]);
                const int cnt = __builtin_amdgcn_readfirstlane(pre_cnt) + 1;  // parking.py:353
                const int frame = __builtin_amdgcn_readfirstlane(pre_frame) + interval_ms;
                const uint32_t tp_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint((float)pre_tp));
                int scen = T2D_SCENARIO_NORMAL, traf = T2D_TRAFFIC_NORMAL;
                if (cfg.max_step > 0 && cnt > cfg.max_step) {
                    scen = T2D_SCENARIO_TIME_EXCEEDED;  // later detectors are not updated (parking.py:366-369)
                } else if (ef & T2D_FLAG_OUT_BOUND) {
                    scen = T2D_SCENARIO_OUT_BOUND;
                } else if (ef & T2D_FLAG_COLLISION_STATIC) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_STATIC;
                } else if (cfg.check_dynamic && (ef & T2D_FLAG_COLLISION_DYNAMIC)) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_DYNAMIC;
                } else if (cfg.check_off_lane && (ef & T2D_FLAG_OFF_LANE)) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_OFF_LANE;
                }
                // ParkingEnv._get_reward (envs/parking.py:148-190) on bit patterns: every value is one of the configuration's
                // floats or the time penalty of the table, rounded to fp32 once, exactly as (float)(double) does below
                uint32_t rb;
                if (traf == T2D_TRAFFIC_COLLISION_STATIC) rb = __float_as_uint(cfg.reward_collision);
                else if (scen == T2D_SCENARIO_TIME_EXCEEDED || scen == T2D_SCENARIO_NO_ACTION) rb = __float_as_uint(cfg.reward_time_exceed);
                else if (scen == T2D_SCENARIO_OUT_BOUND) rb = __float_as_uint(cfg.reward_out_bound);
                else if (scen == T2D_SCENARIO_COMPLETED) rb = __float_as_uint(cfg.reward_completed);
                else if (traf == T2D_TRAFFIC_COLLISION_DYNAMIC || traf == T2D_TRAFFIC_OFF_LANE) rb = __float_as_uint(cfg.reward_collision);
                else rb = cfg.max_step > 0 ? tp_bits : 0u;
                const bool terminated = scen == T2D_SCENARIO_COMPLETED;
                const bool truncated = !terminated && (scen != T2D_SCENARIO_NORMAL || traf != T2D_TRAFFIC_NORMAL);
                const uint32_t st = (uint32_t)scen | (uint32_t)traf << 8 | (uint32_t)terminated << 16 | (uint32_t)truncated << 24;
                const bool done = e_auto_reset && (terminated || truncated);
                e_env_flags[env_s] = env_or;
                e_cnt_step[env_s] = done ? 0 : cnt;
                e_frame_ms[env_s] = done ? 0 : frame;
                if (LOOP) {
                    c_cnt = done ? 0 : cnt;
                    c_frame = done ? 0 : frame;
                }
                ((T2D_GLOBAL uint32_t*)e_status)[env_s] = st;
                ((T2D_GLOBAL uint32_t*)e_reward)[env_s] = rb;
                ((T2D_GLOBAL uint32_t*)e_iou)[env_s] = 0x7fc00000u;   // NaN: not evaluated
                const unsigned long long rec = (unsigned long long)rb | ((unsigned long long)st << 32);
                if (MULTI) ((T2D_GLOBAL unsigned long long*)e_record_ring)[(size_t)((e_record_slot0 + step_k) & (T2D_RECORD_RING - 1)) * (size_t)e_n_env + env_s] = rec;
                else ((T2D_GLOBAL unsigned long long*)e_record)[env_s] = rec;
                if (e_auto_reset) {
                    s_done[env_local] = done;
                    if (PIPE) s_dec[env_local] = done;   // (the integrator wave reads it before it commits the next step)
                    if (done) {  // ParkingEnv.reset: detector state back to the episode start (the counters: above)
                        e_last_valid[env_s] = 0;
                        e_cnt_na[env_s] = 0;
                        e_max_iou[env_s] = -INFINITY;
                        e_min_dist[env_s] = e_snap_min_dist[env_s];
                    }
                }
            }
        }
#endif
    }
    if (!epilogue_done && valid && agent == 0 && !(T2D_PROBE_SKIP & 128)) {
        e_env_flags[env] = s_env_or[env_local];
        if (WITH_STATUS) {
            const int cnt = pre_cnt + 1;  // parking.py:353
            e_cnt_step[env] = cnt;
            e_frame_ms[env] = pre_frame + interval_ms;
            if (LOOP) {
                c_cnt = cnt;
                c_frame = pre_frame + interval_ms;
            }
            const int ego = slot0 + cfg.ego_index;  // workgroup-local lane of the ego
            const uint32_t ef = s_flags[ego];
            int scen = T2D_SCENARIO_NORMAL, traf = T2D_TRAFFIC_NORMAL;
            double iou = 0.0;
            bool has_iou = false;
            const bool ego_obb = s_kind[ego] == T2D_SHAPE_OBB;
            if (cfg.max_step > 0 && cnt > cfg.max_step) {
                scen = T2D_SCENARIO_TIME_EXCEEDED;  // later detectors are not updated (parking.py:366-369)
            } else {
                bool na = false;
                if (IOU && cfg.check_no_action && ego_obb) {  // NoAction.update (no_action.py:41-53)
                    double* last = pv.last_pose + 8 * (size_t)env;
                    int cna = e_cnt_na[env];
                    if (!e_last_valid[env]) {
                        e_last_valid[env] = 1;
                    } else {
                        cna = iou_na > (double)cfg.no_action_iou ? cna + 1 : 0;
                        e_cnt_na[env] = cna;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        last[2 * k] = s_v[2 * k][ego];
                        last[2 * k + 1] = s_v[2 * k + 1][ego];
                    }
                    na = cna > cfg.no_action_max_step;
                }
                if (na) {
                    traf = T2D_TRAFFIC_NO_ACTION_QUIRK;  // parking.py:373 writes ScenarioStatus.NO_ACTION here
                } else if (ef & T2D_FLAG_OUT_BOUND) {
                    scen = T2D_SCENARIO_OUT_BOUND;
                } else if (ef & T2D_FLAG_COLLISION_STATIC) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_STATIC;
                } else if (cfg.check_dynamic && (ef & T2D_FLAG_COLLISION_DYNAMIC)) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_COLLISION_DYNAMIC;
                } else if (cfg.check_off_lane && (ef & T2D_FLAG_OFF_LANE)) {
                    scen = T2D_SCENARIO_FAILED; traf = T2D_TRAFFIC_OFF_LANE;
                } else if (IOU && cfg.check_arrival && pv.target_xy && ego_obb) {  // Arrival.update (arrival.py:42-47)
                    iou = iou_ar;
                    has_iou = true;
                    if (iou >= (double)cfg.arrival_threshold) scen = T2D_SCENARIO_COMPLETED;
                }
            }
            double rd;  // ParkingEnv._get_reward (envs/parking.py:148-190)
            if (traf == T2D_TRAFFIC_COLLISION_STATIC) rd = cfg.reward_collision;
            else if (scen == T2D_SCENARIO_TIME_EXCEEDED || scen == T2D_SCENARIO_NO_ACTION) rd = cfg.reward_time_exceed;
            else if (scen == T2D_SCENARIO_OUT_BOUND) rd = cfg.reward_out_bound;
            else if (scen == T2D_SCENARIO_COMPLETED) rd = cfg.reward_completed;
            else if (traf == T2D_TRAFFIC_COLLISION_DYNAMIC || traf == T2D_TRAFFIC_OFF_LANE) rd = cfg.reward_collision;
            else {
                rd = cfg.max_step > 0 ? (e_time_penalty ? pre_tp : -tanh((double)cnt / (double)cfg.max_step) * (double)cfg.time_penalty_scale) : 0.0;
                if (cfg.shaped_reward) {
                    double mi = e_max_iou[env];
                    double iou_reward = 0.0;
                    if (has_iou) iou_reward = mi == -INFINITY ? iou : iou - mi;
                    rd = rd + iou_reward;
                    if (has_iou) e_max_iou[env] = mi > iou ? mi : iou;
                    if (e_target_c) {
                        const double dx = (double)s_cxy[0][ego] - e_target_c[2 * (size_t)env];
                        const double dy = (double)s_cxy[1][ego] - e_target_c[2 * (size_t)env + 1];
                        const double d = __builtin_sqrt(dx * dx + dy * dy);
                        const double md = e_min_dist[env];
                        if (d < md) {
                            rd += (md - d) * (double)cfg.dist_reward_scale;
                            e_min_dist[env] = d;
                        }
                    }
                }
            }
            const float r = (float)rd;
            e_iou[env] = has_iou ? (float)iou : __builtin_nanf("");
            const bool terminated = scen == T2D_SCENARIO_COMPLETED;
            const bool truncated = !terminated && (scen != T2D_SCENARIO_NORMAL || traf != T2D_TRAFFIC_NORMAL);
            // {scenario, traffic, terminated, truncated}: the four status bytes are also the record's second word
            const uint32_t st = (uint32_t)scen | (uint32_t)traf << 8 | (uint32_t)terminated << 16 | (uint32_t)truncated << 24;
            ((T2D_GLOBAL uint32_t*)e_status)[env] = st;
            e_reward[env] = r;
            // (the record {reward bits, status word} as one 8-byte store)
            const unsigned long long rec = (unsigned long long)__float_as_uint(r) | ((unsigned long long)st << 32);
            if (MULTI) ((T2D_GLOBAL unsigned long long*)e_record_ring)[(size_t)((e_record_slot0 + step_k) & (T2D_RECORD_RING - 1)) * (size_t)e_n_env + env] = rec;
            else ((T2D_GLOBAL unsigned long long*)e_record)[env] = rec;
            if (e_auto_reset) {
                const bool done = terminated || truncated;
                s_done[env_local] = done;
                if (PIPE) s_dec[env_local] = done;   // (the integrator wave reads it before it commits the next step)
                if (done) {  // ParkingEnv.reset: counters and detector state back to the episode start
                    e_cnt_step[env] = 0;
                    e_frame_ms[env] = 0;
                    if (LOOP) c_cnt = c_frame = 0;
                    e_last_valid[env] = 0;
                    e_cnt_na[env] = 0;
                    e_max_iou[env] = -INFINITY;
                    e_min_dist[env] = e_snap_min_dist[env];
                }
            }
        }
    }
    if (LOOP) {   // what this trip stored (a pure-output velocity excepted: only a point mass reads c_vx / c_vy back)
        c_ids = ids; c_x = fx; c_y = fy; c_h = fh; c_v = fv;
    }
    if (WITH_STATUS && e_auto_reset && !PIPE && !(T2D_PROBE_SKIP & 256)) {  // fused vector-env auto-reset: finished envs go back to the snapshot (PIPE: by the integrator wave)
        // (the restore's eighteen pointers are requested here, in one scalar round trip, not with the epilogue's above: 36
        // more scalar registers held through the reduce and the status code pushed the kernel into scalar spills -- two
        // v_readlane / v_writelane per spilled value, ~300 VALU instructions per wave)
        const KernargView rp = late_args();
        auto e_snap0 = as_global(rp->snap[0]);
        auto e_snap1 = as_global(rp->snap[1]);
        auto e_snap2 = as_global(rp->snap[2]);
        auto e_snap3 = as_global(rp->snap[3]);
        auto e_snap4 = as_global(rp->snap[4]);
        auto e_snap5 = as_global(rp->snap[5]);
        auto e_snap_ids = as_global(rp->snap_ids);
        auto e_snap_omega0 = as_global(rp->snap_omega[0]);
        auto e_snap_omega1 = as_global(rp->snap_omega[1]);
        auto e_x = as_global(rp->x);
        auto e_y = as_global(rp->y);
        auto e_heading = as_global(rp->heading);
        auto e_speed = as_global(rp->speed);
        auto e_vx = as_global(rp->vx);
        auto e_vy = as_global(rp->vy);
        auto e_ids = as_global(rp->ids);
        auto e_omega_f = as_global(rp->omega_f);
        auto e_omega_r = as_global(rp->omega_r);
        asm volatile("" : "+s"(e_snap0), "+s"(e_snap1), "+s"(e_snap2), "+s"(e_snap3), "+s"(e_snap4), "+s"(e_snap5), "+s"(e_snap_ids), "+s"(e_snap_omega0), "+s"(e_snap_omega1), "+s"(e_x), "+s"(e_y), "+s"(e_heading), "+s"(e_speed), "+s"(e_vx), "+s"(e_vy), "+s"(e_ids), "+s"(e_omega_f), "+s"(e_omega_r));
        if (log2A <= 6) wave_sync(); else __syncthreads();
        if (valid && s_done[env_local]) {
            // every snapshot value first, then the stores: written as load / store pairs, each pair waits for its own
            // memory round trip (a store may alias the next load as far as the compiler knows) -- seven in a row at
            // the very end of the wave, where nothing is left to hide them
            const float r0 = e_snap0[idx], r1 = e_snap1[idx], r2 = e_snap2[idx], r3 = e_snap3[idx];
            const float r4 = e_snap4[idx], r5 = e_snap5[idx];
            const uint32_t rid = e_snap_ids[idx];
            const bool drift = e_snap_omega0 != nullptr;   // SingleTrackDrift wheel speeds (only with a drift type in the table)
            float w0 = 0.f, w1 = 0.f;
            if (drift) {
                w0 = e_snap_omega0[idx];
                w1 = e_snap_omega1[idx];
            }
            e_x[idx] = r0;
            e_y[idx] = r1;
            e_heading[idx] = r2;
            e_speed[idx] = r3;
            e_vx[idx] = r4;
            e_vy[idx] = r5;
            e_ids[idx] = rid;
            if (LOOP) {
                c_x = r0; c_y = r1; c_h = r2; c_v = r3; c_vx = r4; c_vy = r5; c_ids = rid;
            }
            if (drift) {
                e_omega_f[idx] = w0;
                e_omega_r[idx] = w1;
            }
        }
    }
    }   // (wave 0 of a SPLIT workgroup)
    T2D_MARK(12);
    if (!LOOP) break;
    // LOOP: nothing is read back from memory (see `carried`); what the next trip clears in LDS is the wave's own when an env
    // fits a wave, else the workgroup meets first
    if (log2A <= 6) wave_sync(); else __syncthreads();
    if constexpr (PIPE == 2) {
        if (!role_b) s_flags[tid] = 0;   // (see the top of the trip)
    }
    if constexpr (PIPE != 0) {
        if (!role_b) pipe_post(&s_seq_e[tid >> 6], (uint32_t)step_k + 1u);   // this step's verdicts are in s_dec
    }
    if (++step_k >= step_end) break;
    }
    if (CHAIN) {   // this step of these envs is complete: every store above is in the L2 before the word moves
#ifdef T2D_TIMING
        t_prev_ = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ptid == 0) {
            unsigned long long w = chain_word(pv.chain_base + (uint32_t)step_k + (LOOP ? 0u : 1u));   // (LOOP: step_k = one past its last step)
            // test hook (t2d_debug_chain_fault): workgroup 1 hands its step 1 over with a foreign XCC id (1) / not at all (2);
            // 3: it hands its step 0 over with a foreign XCC id -- the failure is then posted while the fragment's first
            // step is still being dispatched on a grid larger than the device holds
#ifdef T2D_DEBUG_HOOKS   // (libt2d_hip_debug.so only; the product posts the word as it is)
            const uint32_t cf = pv.chain_fault;
            const uint32_t fault = unit != 1 ? 0u : (cf == 3u ? (step_k == 0 ? 1u : 0u) : (step_k == 1 ? cf : 0u));
            if (fault & 1u) w ^= 1ull << 32;
            if (!(fault & 2u))
#endif
            __hip_atomic_store(&pv.chain_done[unit], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef T2D_TIMING
        if constexpr (CHAIN) {
            T2D_MARK(19);
            if (lane == 0) {
                pv.dbg[wave_slot_ + 17] = __builtin_amdgcn_s_memrealtime();
                pv.dbg[wave_slot_ + 20] = t_prev_;
            }
        }
#endif
    }
#undef pv
#undef cfg
}

}  // namespace

// Every launch of collide_kernel goes through this macro: the debug library (-DT2D_DEBUG_HOOKS) notes WHICH instantiation the
// last launch took (t2d_debug_last_step_kernel), so that tests/test_gpu_forms.py can hold the list of instantiations in this
// file against what the test suite's recipes actually reach (the table of forms).
#ifdef T2D_DEBUG_HOOKS
const char* g_last_collide_form = "";
#define T2D_NOTE_FORM(str) (g_last_collide_form = str)
#else
#define T2D_NOTE_FORM(str) ((void)0)
#endif
#define T2D_UNPAREN(...) __VA_ARGS__
#define T2D_LAUNCH_COLLIDE(ARGS, grid_, block_)                                                                              \
    do {                                                                                                                     \
        T2D_NOTE_FORM(#ARGS);                                                                                                \
        hipLaunchKernelGGL((collide_kernel<T2D_UNPAREN ARGS>), grid_, block_, dyn, s, v, cfg, interval_ms, log2A);           \
    } while (0)

// resident workgroups per CU of the fused step kernel with this pool's geometry record (the metric scenes are sized
// for 4: one wave-round of 1024 workgroups on 256 CUs) and the LDS bytes per workgroup
hipError_t step_occupancy(const PoolView& v, int* blocks_per_cu, size_t* lds_bytes) {
    int log2A = 1;
    while ((1 << log2A) < v.A) ++log2A;
    const int block = v.geo_layout.epb << log2A;
    const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
    hipFuncAttributes fa{};
    hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&collide_kernel<true, 1, false>));
    if (e != hipSuccess) return e;
    *lds_bytes = fa.sharedSizeBytes + dyn;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, collide_kernel<true, 1, false>, block, dyn);
}

// SPLIT (one env per workgroup, its event stages on four waves): envs of 33..64 participants, workgroups of the full
// 256 threads, no IoU events, and few enough envs for every workgroup of a step to be resident (four per CU)
bool split_eligible(const PoolView& v, const t2d_status_config& cfg, int log2A, int device_cus) {
    // ... and only where there is something to run side by side: envs with static obstacles next to their lanes (measured:
    // roundabout / intersection pools gain 7-10 %; a highway pool -- pairs and a few lane rectangles, nothing else -- loses 9 %
    // to the extra barriers and the poses derived four times)
    return log2A == 6 && !(cfg.check_no_action || cfg.check_arrival) && device_cus > 0 &&
           v.n_env <= 4 * device_cus && !v.wgmap && v.geo && v.geo_layout.has[0] && v.geo_layout.has[1];
}

hipError_t launch_step_chain(const PoolView& v, const t2d_status_config& cfg, int interval_ms, int variant, int n_steps,
                             hipStream_t s) {
    int log2A = 1;
    while ((1 << log2A) < v.A) ++log2A;
    const int EPB = v.geo_layout.epb;
    if (cfg.check_no_action || cfg.check_arrival) return hipErrorInvalidValue;   // (see below)
    if (v.split_step) {   // small pool of 64-agent envs: one env per workgroup, chained per env
        const int padded = (v.n_env + 7) & ~7;
        const dim3 grid(padded, n_steps), block(kBlock);
        const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
        if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, true, false, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, true, false, true), grid, block);
        return hipGetLastError();
    }
#ifdef T2D_EXPERIMENTS   // (measured in round 6 and not shipped: profiles/r06_ab_chain_depth.txt)
    if (v.chain_k > 1) {   // the chained form, chain_k steps per workgroup: grid (env sets, n_steps / chain_k)
        if (log2A > 6 || v.idm_rows || n_steps % v.chain_k) return hipErrorInvalidValue;
        const int real = (v.n_env + EPB - 1) / EPB, padded = (real + 7) & ~7;
        const dim3 grid(padded, n_steps / v.chain_k), block(EPB << log2A);
        const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
        if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, true, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, true, true), grid, block);
        return hipGetLastError();
    }
#endif
    if (v.loop_steps > 0 && v.pipe_step) {   // ... with a second set of waves that integrates a step ahead
        const dim3 grid((v.n_env + EPB - 1) / EPB), block(2 * (EPB << log2A));
        const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
        if (log2A > 6) return hipErrorInvalidValue;
        if (v.pipe_step == 2) {   // ... and a third set that takes the lane stage
            const dim3 block3(3 * (EPB << log2A));
            if (block3.x > 1024) return hipErrorInvalidValue;
            if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, true, false, 2), grid, block3);
            else T2D_LAUNCH_COLLIDE((true, 1, false, false, true, false, 2), grid, block3);
            return hipGetLastError();
        }
        if (v.idm_rows) {   // installed IDM controllers: run by the integrator waves
            if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, true, false, 1, true), grid, block);
            else T2D_LAUNCH_COLLIDE((true, 1, false, false, true, false, 1, true), grid, block);
            return hipGetLastError();
        }
        if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, true, false, 1), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, false, true, false, 1), grid, block);
        return hipGetLastError();
    }
    if (v.loop_steps > 0) {   // small pool: every workgroup resident, each walks through the steps itself
        const dim3 grid((v.n_env + EPB - 1) / EPB), block(EPB << log2A);
        const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
        if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, false, true), grid, block);
        return hipGetLastError();
    }
    // x extent rounded up to a multiple of 8: workgroup ids go round the 8 XCDs, so step k + 1 of a set of envs then runs on
    // the XCD that ran their step k and finds their state in that XCD's L2 (the padding workgroups only move their counter)
    const int real = (v.n_env + EPB - 1) / EPB, padded = (real + 7) & ~7;
    const dim3 grid(padded, n_steps), block(EPB << log2A);
    const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
    // (pools with IoU events keep per-env history -- last pose, counters -- that the epilogue reads through plain pointers:
    // the caller steps those one launch at a time)
    if (cfg.check_no_action || cfg.check_arrival) return hipErrorInvalidValue;
    if (v.idm_rows) {   // installed IDM controllers: run by every workgroup ahead of its integrator
        if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, true, false, false, 0, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, true, false, false, 0, true), grid, block);
        return hipGetLastError();
    }
    if (variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, true), grid, block);
    else T2D_LAUNCH_COLLIDE((true, 1, false, true), grid, block);
    return hipGetLastError();
}

hipError_t launch_collide(const PoolView& v, const t2d_status_config& cfg, bool with_status,
                          int interval_ms, int fuse_variant, hipStream_t s) {
    int log2A = 1;   // at least two lanes per env (see log2_pad in t2d_api.hip)
    while ((1 << log2A) < v.A) ++log2A;
    const int EPB = v.geo_layout.epb;
    const dim3 grid((v.n_env + EPB - 1) / EPB), block(EPB << log2A);
    const size_t dyn = v.geo ? (size_t)v.geo_layout.stride * 4 : 0;
    const bool iou = cfg.check_no_action || cfg.check_arrival;
    if (fuse_variant >= 0 && v.split_step) {
        const dim3 sgrid(v.n_env), sblock(kBlock);
        if (fuse_variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, false, true), sgrid, sblock);
        else T2D_LAUNCH_COLLIDE((true, 1, false, false, false, true), sgrid, sblock);
        return hipGetLastError();
    }
    if (fuse_variant >= 0 && v.idm_rows && !iou) {   // installed IDM controllers, run ahead of the integrator
        if (fuse_variant == 0) T2D_LAUNCH_COLLIDE((true, 0, false, false, false, false, 0, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, 1, false, false, false, false, 0, true), grid, block);
        return hipGetLastError();
    }
    if (fuse_variant >= 0) {  // the fused step always runs the status epilogue
        if (fuse_variant == 0) {
            if (iou) T2D_LAUNCH_COLLIDE((true, 0, true), grid, block);
            else T2D_LAUNCH_COLLIDE((true, 0, false), grid, block);
        } else {
            if (iou) T2D_LAUNCH_COLLIDE((true, 1, true), grid, block);
            else T2D_LAUNCH_COLLIDE((true, 1, false), grid, block);
        }
    } else if (with_status) {
        if (iou) T2D_LAUNCH_COLLIDE((true, -1, true), grid, block);
        else T2D_LAUNCH_COLLIDE((true, -1, false), grid, block);
    } else {
        T2D_LAUNCH_COLLIDE((false, -1, false), grid, block);
    }
    return hipGetLastError();
}

#ifdef T2D_DEBUG_HOOKS
const char* last_collide_form() { return g_last_collide_form; }
#endif

}  // namespace t2d
