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
    records = torch.zeros((2, n, 2), dtype=torch.int32)   # stands in for the pool's T2D_F_RECORD field
    g = D.ResultGather(records, world)
    for step in range(3):
        # what the step kernel writes: reward bits + status word of the local envs, half = step & 1
        env = torch.arange(lo, hi, dtype=torch.float32)
        reward = -0.001 * (step + 1) - env
        status = torch.stack([(torch.arange(lo, hi) % 6 + 1).to(torch.uint8),
                              torch.full((n,), 1 + step, dtype=torch.uint8),
                              torch.zeros(n, dtype=torch.uint8),
                              (torch.arange(lo, hi) % 2).to(torch.uint8)], 1)
        records[step & 1].copy_(D.pack_record(reward, status))
        k = g.launch(step)
        records[(step + 1) & 1].fill_(-1)  # the next step overwrites the OTHER half while this gather flies
        rw, st = g.result(k)
        np.save(os.path.join(out_dir, f"r{rank}_s{step}_rw.npy"), rw.numpy())
        np.save(os.path.join(out_dir, f"r{rank}_s{step}_st.npy"), st.numpy())
    g.wait()
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


def test_pack_unpack_roundtrip():
    from tactics2d_amd.dist import pack_record, unpack_record
    rw = torch.tensor([0.5, -5.0, -0.00037], dtype=torch.float32)
    st = torch.tensor([[1, 1, 0, 0], [6, 3, 0, 1], [2, 1, 1, 0]], dtype=torch.uint8)
    rec = pack_record(rw, st)
    assert rec.shape == (3, 2) and rec.dtype == torch.int32 and rec.element_size() * rec.shape[1] == 8
    rw2, st2 = unpack_record(rec)
    assert torch.equal(rw, rw2) and torch.equal(st, st2)
