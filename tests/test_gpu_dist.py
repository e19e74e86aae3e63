"""Multi-GPU path on a one-GPU box: the native result gather (t2d_gather: RCCL opened by the library, world of one)
and two ranks that each step a REAL pool holding their shard of the environments and exchange the records (gloo, both
ranks on device 0 -- RCCL refuses two ranks on one GPU), against a single pool that owns every environment."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N_ENV, A, STEPS, EVERY = 48, 64, 128, 16


def _scene():
    from tactics2d_amd import scenarios as S
    return S.mixed(N_ENV, A, seed=12)


def _actions(sc):
    rng = np.random.default_rng(99)
    return [sc.sample_actions(rng) for _ in range(STEPS)]


def _single_pool_records(sc, acts):
    """reward / status of every step from ONE pool owning all environments (downloaded step by step)."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_auto_reset(True)
    out = []
    for a0, a1 in acts:
        pool.set_actions(a0, a1)
        pool.step(100)
        out.append((pool.download(L.F_REWARD), pool.download(L.F_STATUS)))
    pool.close()
    return out


@pytest.mark.parametrize("with_rccl", [False, True])
def test_native_gather_reads_the_record_ring_in_place(with_rccl):
    """t2d_gather on a world of one: every fragment of 16 steps arrives complete while later steps keep overwriting the
    ring (64 steps = the 32-slot ring twice), with and without an RCCL communicator behind it."""
    from tactics2d_amd.dist import NativeGather
    from tactics2d_amd.pool import ParticipantPool
    sc = _scene()
    acts = _actions(sc)
    want = _single_pool_records(sc, acts)
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_auto_reset(True)
    if with_rccl:
        pool.comm_init(pool.comm_unique_id(), 0, 1)       # ncclCommInitRank, one rank: dlopen + the real collective
    else:
        NativeGather.bootstrap(pool, 0, 1)
    g = NativeGather(pool, 1, every=EVERY, device="cuda")
    st = torch.cuda.Stream()
    got = {}
    pending = None
    for t, (a0, a1) in enumerate(acts):
        pool.set_actions(a0, a1)
        pool.step(100, st.cuda_stream)
        k = g.launch(stream=st.cuda_stream)
        if pending is not None and k is None and (t + 1) % EVERY == 3:   # read a fragment while later steps are in flight
            kk, t_last = pending
            for j in range(EVERY):
                rw, s = g.result(kk, j)
                got[t_last - EVERY + 1 + j] = (rw.cpu().numpy(), s.cpu().numpy())
            pending = None
        if k is not None:
            pending = (k, t)
    kk, t_last = pending
    for j in range(EVERY):
        rw, s = g.result(kk, j)
        got[t_last - EVERY + 1 + j] = (rw.cpu().numpy(), s.cpu().numpy())
    pool.close()
    assert sorted(got) == list(range(STEPS))
    for t in range(STEPS):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1]), t
    assert any(w[1][:, 3].any() for w in want)            # some episodes ended: the records are not trivial


def _device_action_ring(sc, acts, dev="cuda"):
    a0 = torch.from_numpy(np.stack([a[0] for a in acts])).to(dev).contiguous()
    a1 = torch.from_numpy(np.stack([a[1] for a in acts])).to(dev).contiguous()
    return a0, a1


@pytest.mark.parametrize("chained", [False, True])
def test_gathers_overlap_the_steps_with_no_host_synchronisation(chained):
    """Actions bound once in device memory, 64 steps and their four gathers enqueued on one stream with no host wait in
    between (so a step really runs while a gather is in flight, and the slot-event wait is what protects the ring); the
    fragments are copied out stream-ordered and compared at the end with a single pool stepped one call at a time."""
    from tactics2d_amd.dist import NativeGather
    from tactics2d_amd.pool import ParticipantPool
    sc = _scene()
    acts = _actions(sc)
    want = _single_pool_records(sc, acts)
    a0, a1 = _device_action_ring(sc, acts)
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_auto_reset(True)
    pool.comm_init(pool.comm_unique_id(), 0, 1)
    assert pool.comm_info() == (True, 1, 0)               # read back from the RCCL communicator itself
    g = NativeGather(pool, 1, every=EVERY, device="cuda")
    st = torch.cuda.Stream()
    keep = torch.zeros((STEPS // EVERY, EVERY, sc.n_env, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for f in range(STEPS // EVERY):
        base = f * EVERY
        if chained:
            pool.bind_actions(a0.data_ptr() + 4 * sc.n * base, a1.data_ptr() + 4 * sc.n * base)
            pool.step_n(EVERY, 100, sc.n, st.cuda_stream)
        else:
            for j in range(EVERY):
                pool.bind_actions(a0.data_ptr() + 4 * sc.n * (base + j), a1.data_ptr() + 4 * sc.n * (base + j))
                pool.step(100, st.cuda_stream)
        k = g.launch(stream=st.cuda_stream)
        assert k == f % 2
        pool.gather_wait(st.cuda_stream, block_host=False)   # the STREAM waits for the gather, the host does not
        with torch.cuda.stream(st):
            keep[f].copy_(g.out[k][0])
    st.synchronize()
    assert pool.step_count() == STEPS
    pool.close()
    from tactics2d_amd.dist import unpack_record
    for t in range(STEPS):
        rw, s = unpack_record(keep[t // EVERY, t % EVERY])
        assert np.array_equal(rw.cpu().numpy(), want[t][0]) and np.array_equal(s.cpu().numpy(), want[t][1]), t


def test_a_step_waits_for_the_gather_that_still_reads_its_record_slot():
    """The record ring has RING slots.  A gather of steps 0..15 is held back on the pool's gather stream (30 ms behind an idle
    kernel: a slow peer), while RING further steps are enqueued at once -- steps RING..RING+15 write the very slots that gather
    has not read yet.  t2d_step must make the step stream wait for it: the fragment arrives intact.  (Without the wait the
    fragment would hold the records of steps 32..47.)"""
    from tactics2d_amd.dist import NativeGather, unpack_record
    from tactics2d_amd.pool import ParticipantPool
    sc = _scene()
    acts = _actions(sc)
    want = _single_pool_records(sc, acts)
    a0, a1 = _device_action_ring(sc, acts)
    from tactics2d_amd import debug as D     # (the gather delay is a hook of libt2d_hip_debug.so: include/t2d_debug.h)
    pool = D.pool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_auto_reset(True)
    pool.comm_init(pool.comm_unique_id(), 0, 1)
    g = NativeGather(pool, 1, every=EVERY, device="cuda")
    st = torch.cuda.Stream()

    def steps(lo, hi):
        for t in range(lo, hi):
            pool.bind_actions(a0.data_ptr() + 4 * sc.n * t, a1.data_ptr() + 4 * sc.n * t)
            pool.step(100, st.cuda_stream)
    torch.cuda.synchronize()
    steps(0, EVERY)
    D.delay_gather(pool, 30000)
    k = g.launch(stream=st.cuda_stream)
    from tactics2d_amd import layout as L
    steps(EVERY, L.RECORD_RING + EVERY)                       # wraps the ring onto slots 0..15 -- must wait inside t2d_step
    rw0, s0 = [], []
    for j in range(EVERY):
        rw, s = g.result(k, j)
        rw0.append(rw.cpu().numpy()); s0.append(s.cpu().numpy())
    st.synchronize()
    pool.close()
    for j in range(EVERY):
        assert np.array_equal(rw0[j], want[j][0]) and np.array_equal(s0[j], want[j][1]), j
    assert not all(np.array_equal(want[j][0], want[L.RECORD_RING + j][0]) for j in range(EVERY))   # the two candidates differ


def test_gather_argument_checks():
    from tactics2d_amd import _ffi
    from tactics2d_amd.pool import ParticipantPool
    sc = _scene()
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    out = torch.empty((1, 8, sc.n_env, 2), dtype=torch.int32, device="cuda")
    with pytest.raises(_ffi.T2DError):
        pool.gather(8, out.data_ptr())                    # no steps taken yet
    a0, a1 = sc.sample_actions(np.random.default_rng(0))
    pool.set_actions(a0, a1)
    for _ in range(3):
        pool.step(100)
    with pytest.raises(_ffi.T2DError):
        pool.gather(3, out.data_ptr())                    # 3 does not divide the ring
    with pytest.raises(_ffi.T2DError):
        pool.gather(2, out.data_ptr())                    # 3 steps taken: not a multiple of 2
    pool.step(100)
    pool.gather(4, out.data_ptr())
    pool.gather_wait(block_host=True)
    with pytest.raises(_ffi.T2DError):
        pool.comm_init(None, 0, 2)                        # a world of two needs the communicator id
    pool.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tactics2d_amd import dist as D, layout as L
    from tactics2d_amd.pool import ParticipantPool
    D.init_process_group("gloo")
    torch.cuda.set_device(0)                              # both ranks on the one GPU of the box
    sc = _scene()
    acts = _actions(sc)
    lo, hi = D.shard_range(sc.n_env, rank, world)
    part = sc.shard(lo, hi)
    pool = ParticipantPool(part.n_env, part.A, 0)
    part.load(pool)
    pool.set_auto_reset(True)
    rec = torch.as_tensor(pool.device_array(L.F_RECORD), device="cuda:0").view(torch.int32)
    g = D.ResultGather(rec, world, every=EVERY)
    rows = []
    for t, (a0, a1) in enumerate(acts):
        pool.set_actions(a0[lo * A:hi * A], a1[lo * A:hi * A])
        pool.step(100)
        pool.sync()                                       # gloo reads the record tensor from the host side
        k = g.launch(t)
        if k is not None:
            for j in range(EVERY):
                rw, s = g.result(k, j)
                rows.append((rw.cpu().numpy().copy(), s.cpu().numpy().copy()))
    np.save(os.path.join(out_dir, f"rank{rank}_reward.npy"), np.stack([r[0] for r in rows]))
    np.save(os.path.join(out_dir, f"rank{rank}_status.npy"), np.stack([r[1] for r in rows]))
    pool.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_real_pools_gather_what_one_pool_computes(tmp_path):
    """N > 1 with pools behind it: two processes, each stepping its contiguous shard of scenarios.mixed through its own
    pool for 32 steps, gather the per-env records fragment by fragment; every rank must end up with exactly the
    records a single pool owning all environments produces (envs never interact, shards are contiguous, rank-major
    order = env order)."""
    import torch.multiprocessing as mp
    sc = _scene()
    want = _single_pool_records(sc, _actions(sc))
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        rw = np.load(tmp_path / f"rank{rank}_reward.npy"); st = np.load(tmp_path / f"rank{rank}_status.npy")
        assert rw.shape == (STEPS, N_ENV) and st.shape == (STEPS, N_ENV, 4)
        for t in range(STEPS):
            assert np.array_equal(rw[t], want[t][0]) and np.array_equal(st[t], want[t][1]), (rank, t)


def test_bench_line_holds_a_gather_inside_the_timed_region_whatever_the_step_count():
    """bench.py with the gather forced on one rank (what N > 1 runs per rank), at a driver-like `--steps 20 --warmup 5`: the
    cadence is cut to what fits the run and the fragments end where the pool's step count reaches a multiple of it, so the
    timed region holds its gather (and the wait for it) -- not a region the collective happens to fall outside of."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, T2D_FORCE_GATHER="1")
    for extra, want_every, want_gathers in ((["--steps", "20", "--warmup", "5"], 16, 1), (["--steps", "64", "--warmup", "7"], 32, 2),
                                            (["--steps", "20", "--warmup", "5", "--mode", "step"], 16, 1)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--envs", "512", "--no-cpu-baseline", "--no-configs",
                              "--no-next-rows", "--no-alternates", "--no-profile", "--clock-warm", "0"] + extra,
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-1500:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        g = line["gather"]
        assert g["every"] == want_every and g["gathers_in_timed_region"] >= want_gathers, g


def test_two_rank_bench_line_carries_the_weak_value_and_the_fixed_total_cases():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE device with
    gloo (T2D_DIST_BACKEND / T2D_FORCE_DEVICE: the records travel through torch.distributed, everything else is the N > 1 code
    path): the line keeps the weak-scaling contract (`scaling: "weak"`, value over all ranks) and carries, under `strong`, the
    fixed-total cases BASELINE.json names -- the metric at 4096 x 64 in total, cfg4 2048 x 32, cfg5 8192 x 64 -- each with the
    gather inside its timed region."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, T2D_DIST_BACKEND="gloo", T2D_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--envs", "1024", "--no-profile", "--clock-warm", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2500:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "weak" and line["n_gpus"] == 2 and line["gather"]["gathers_in_timed_region"] >= 1
    st = line["strong"]
    assert set(st) >= {"metric_4096x64", "cfg4_2048x32", "cfg5_8192x64", "note"}, st
    for key, per in (("metric_4096x64", 2048), ("cfg4_2048x32", 1024), ("cfg5_8192x64", 4096)):
        c = st[key]
        assert c["envs_per_gpu"] == per and c["value"] > 1e7 and c["gather_every"] == 16 and c["gather_native"] is False, c
        assert abs(c["value"] - c["total_envs"] * c["participants_per_env"] / (c["us_per_step"] * 1e-6)) < 1e-3 * c["value"]


def _rank_native(rank, world, port, out_dir):
    """one process per GPU: the library's own RCCL communicator (t2d_comm_init with a real unique id), steps and gathers on
    one stream with no host wait in between -- what `bench.py --gpus N` runs"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tactics2d_amd import dist as D
    from tactics2d_amd.pool import ParticipantPool
    D.init_process_group("nccl")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    sc = _scene()
    acts = _actions(sc)[:64]
    lo, hi = D.shard_range(sc.n_env, rank, world)
    part = sc.shard(lo, hi)
    pool = ParticipantPool(part.n_env, part.A, rank)
    part.load(pool)
    pool.set_auto_reset(True)
    D.NativeGather.bootstrap(pool, rank, world)
    native, cw, cr = pool.comm_info()
    assert native and cw == world and cr == rank, (native, cw, cr)
    g = D.NativeGather(pool, world, every=EVERY, device=dev)
    a0 = torch.from_numpy(np.stack([a[0][lo * A:hi * A] for a in acts])).to(dev).contiguous()
    a1 = torch.from_numpy(np.stack([a[1][lo * A:hi * A] for a in acts])).to(dev).contiguous()
    st = torch.cuda.Stream(device=dev)
    rows = []
    n = part.n
    for t in range(len(acts)):
        pool.bind_actions(a0.data_ptr() + 4 * n * t, a1.data_ptr() + 4 * n * t)
        pool.step(100, st.cuda_stream)
        k = g.launch(None, st.cuda_stream)
        if k is not None:
            for j in range(EVERY):
                rw, s = g.result(k, j)
                rows.append((rw.cpu().numpy().copy(), s.cpu().numpy().copy()))
    np.save(os.path.join(out_dir, f"nrank{rank}_reward.npy"), np.stack([r[0] for r in rows]))
    np.save(os.path.join(out_dir, f"nrank{rank}_status.npy"), np.stack([r[1] for r in rows]))
    pool.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_native_gather_two_gpus(tmp_path):
    """The first execution of t2d_gather with more than one rank (round-3 review): two processes, one GPU each, the library's
    own RCCL communicator created from a real unique id, 64 steps with a gather of the 8-byte env records every 16 -- every
    rank must hold exactly the records one pool owning all the environments produces, rank-major = env order
    (independence of the envs: traffic/scenario_manager.py:52-61).  Skipped on the one-GPU boxes of the round's GPU tier; the
    8-GPU node of the scaling run takes it."""
    import torch.multiprocessing as mp
    sc = _scene()
    want = _single_pool_records(sc, _actions(sc)[:64])
    world = 2
    mp.spawn(_rank_native, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        rw = np.load(tmp_path / f"nrank{rank}_reward.npy"); st = np.load(tmp_path / f"nrank{rank}_status.npy")
        assert rw.shape == (64, N_ENV) and st.shape == (64, N_ENV, 4)
        for t in range(64):
            assert np.array_equal(rw[t], want[t][0]) and np.array_equal(st[t], want[t][1]), (rank, t)
