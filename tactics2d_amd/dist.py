"""Multi-GPU plumbing: one process per GPU, environments sharded in contiguous blocks, and the
only collective on the path -- an all-gather of the 8-byte per-env result record {reward f32,
scenario_status u8, traffic_status u8, terminated u8, truncated u8}.

Environments never interact (traffic/scenario_manager.py:52-61: one ScenarioManager owns one
scene), so there is no data-path exchange: every rank steps its own pool.  torch.distributed is
plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np


def env_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults 0, 0, 1)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(n_env_total, rank, world):
    """Contiguous block sharding: rank r owns envs [r*E/G, (r+1)*E/G) (remainder to low ranks)."""
    base, rem = divmod(n_env_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend=None):
    import torch.distributed as dist
    rank, local_rank, world = env_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world)
    return rank, local_rank, world


def pack_record(reward, status):
    """reward f32[E], status u8[E,4] (torch tensors, same device) -> int32[E,2] record."""
    import torch
    rec = torch.empty((reward.shape[0], 2), dtype=torch.int32, device=reward.device)
    rec[:, 0] = reward.view(torch.int32)
    rec[:, 1] = status.contiguous().view(torch.int32).reshape(-1)
    return rec


def unpack_record(rec):
    """int32[E,2] -> (reward f32[E], status u8[E,4])."""
    import torch
    reward = rec[:, 0].contiguous().view(torch.float32)
    status = rec[:, 1].contiguous().view(torch.uint8).reshape(-1, 4)
    return reward, status


class ResultGather:
    """Asynchronous all-gather of the per-env result records.

    The step kernel itself writes the 8-byte records into a double-buffered pool field
    (T2D_F_RECORD, half = step parity), so the collective reads them in place: no packing kernels,
    and step k+1 may run while the gather of step k is still in flight.  `records` is an int32
    tensor view [2, E, 2] of that field (a CPU tensor in the gloo tests)."""

    def __init__(self, records, world):
        import torch
        self.world = world
        self.records = records
        n = records.shape[1]
        self.out = [torch.empty((world * n, 2), dtype=torch.int32, device=records.device) for _ in range(2)]
        self.work = [None, None]

    def launch(self, step):
        """Start gathering the records of (0-based) step `step`; returns the buffer index."""
        import torch.distributed as dist
        k = step & 1
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        if self.world > 1:
            self.work[k] = dist.all_gather_into_tensor(self.out[k], self.records[k], async_op=True)
        else:
            self.out[k].copy_(self.records[k])
        return k

    def wait(self, k=None):
        for i in ([k] if k is not None else [0, 1]):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def result(self, k):
        """(reward f32[world*E], status u8[world*E,4]) of buffer k, rank-major env order."""
        self.wait(k)
        return unpack_record(self.out[k])
