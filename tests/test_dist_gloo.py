"""The N > 1 path on CPU: two gloo ranks, each owning a shard of the environments, exchange the
8-byte per-env result records through tactics2d_amd.dist.ResultGather (the bench's collective)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_env_total, out_dir):
    import torch.distributed as dist
    from tactics2d_amd import dist as D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(n_env_total, rank, world)
    n = hi - lo
    ring = 16
    records = torch.zeros((ring, n, 2), dtype=torch.int32)   # stands in for the pool's T2D_F_RECORD field
    g = D.ResultGather(records, world)
    for step in range(3):
        # what the step kernel writes: reward bits + status word of the local envs, half = step & 1
        env = torch.arange(lo, hi, dtype=torch.float32)
        reward = -0.001 * (step + 1) - env
        status = torch.stack([(torch.arange(lo, hi) % 6 + 1).to(torch.uint8),
                              torch.full((n,), 1 + step, dtype=torch.uint8),
                              torch.zeros(n, dtype=torch.uint8),
                              (torch.arange(lo, hi) % 2).to(torch.uint8)], 1)
        records[step % ring].copy_(D.pack_record(reward, status))
        k = g.launch(step)
        records[(step + 1) % ring].fill_(-1)  # the next step overwrites the NEXT slot while this gather flies
        rw, st = g.result(k)
        np.save(os.path.join(out_dir, f"r{rank}_s{step}_rw.npy"), rw.numpy())
        np.save(os.path.join(out_dir, f"r{rank}_s{step}_st.npy"), st.numpy())
    g.wait()
    # fragments of 4 steps in one message (the bench's setting is 8): steps 0..11, three fragments
    g4 = D.ResultGather(records, world, every=4)
    frag = []
    for step in range(12):
        env = torch.arange(lo, hi, dtype=torch.float32)
        status = torch.zeros((n, 4), dtype=torch.uint8); status[:, 1] = step
        records[step % ring].copy_(D.pack_record(-env - 100.0 * step, status))
        k = g4.launch(step)
        assert (k is None) == ((step + 1) % 4 != 0)
        if k is not None:
            for j in range(4):
                rw, st = g4.result(k, j)
                frag.append((step - 3 + j, rw.numpy().copy(), st.numpy().copy()))
    np.save(os.path.join(out_dir, f"r{rank}_frag_rw.npy"), np.stack([f[1] for f in frag]))
    np.save(os.path.join(out_dir, f"r{rank}_frag_step.npy"), np.array([f[0] for f in frag]))
    np.save(os.path.join(out_dir, f"r{rank}_frag_st.npy"), np.stack([f[2][:, 1] for f in frag]))
    g4.wait()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_result_gather(tmp_path):
    import torch.multiprocessing as mp
    world, total = 2, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    for step in range(3):
        env = np.arange(total, dtype=np.float32)
        want_rw = (-0.001 * (step + 1) - env).astype(np.float32)
        for rank in range(world):
            rw = np.load(tmp_path / f"r{rank}_s{step}_rw.npy"); st = np.load(tmp_path / f"r{rank}_s{step}_st.npy")
            assert rw.shape == (total,) and st.shape == (total, 4)
            assert np.array_equal(rw, want_rw)                 # rank-major == env order for equal shards
            assert np.array_equal(st[:, 0], np.arange(total) % 6 + 1)
            assert (st[:, 1] == 1 + step).all() and np.array_equal(st[:, 3], np.arange(total) % 2)
    env = np.arange(total, dtype=np.float32)
    for rank in range(world):
        steps = np.load(tmp_path / f"r{rank}_frag_step.npy")
        rw = np.load(tmp_path / f"r{rank}_frag_rw.npy"); st1 = np.load(tmp_path / f"r{rank}_frag_st.npy")
        assert list(steps) == list(range(12))
        for i, sp in enumerate(steps):
            assert np.array_equal(rw[i], (-env - 100.0 * sp).astype(np.float32)) and (st1[i] == sp).all()


def test_result_gather_rejects_bad_fragment_sizes():
    from tactics2d_amd.dist import ResultGather
    rec = torch.zeros((16, 4, 2), dtype=torch.int32)
    for bad in (0, 3, 16):
        with pytest.raises(ValueError):
            ResultGather(rec, 1, every=bad)
    g = ResultGather(rec, 1, every=8)
    rec[3, :, 0] = 7
    assert g.launch(6) is None and g.launch(7) == 0
    assert (g.out[0][0, 3, :, 0] == 7).all()


def test_pack_unpack_roundtrip():
    from tactics2d_amd.dist import pack_record, unpack_record
    rw = torch.tensor([0.5, -5.0, -0.00037], dtype=torch.float32)
    st = torch.tensor([[1, 1, 0, 0], [6, 3, 0, 1], [2, 1, 1, 0]], dtype=torch.uint8)
    rec = pack_record(rw, st)
    assert rec.shape == (3, 2) and rec.dtype == torch.int32 and rec.element_size() * rec.shape[1] == 8
    rw2, st2 = unpack_record(rec)
    assert torch.equal(rw, rw2) and torch.equal(st, st2)
