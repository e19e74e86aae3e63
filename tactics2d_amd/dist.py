"""Multi-GPU plumbing: one process per GPU, environments sharded in contiguous blocks, and the
only collective on the path -- an all-gather of the 8-byte per-env result record {reward f32,
scenario_status u8, traffic_status u8, terminated u8, truncated u8}.

Environments never interact (traffic/scenario_manager.py:52-61: one ScenarioManager owns one
scene), so there is no data-path exchange: every rank steps its own pool.  torch.distributed is
plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os



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

    The step kernel itself writes the 8-byte records into a ring of slots in the pool (T2D_F_RECORD, slot =
    step number % ring), so the collective reads them in place: no packing kernels.  `every` = K ships the
    records of K consecutive steps in ONE message (a rollout fragment): a collective costs ~10 us of launch /
    stream-event overhead per call however small it is, which per step would be a third of the step itself;
    K must divide the ring and leave at least two fragments in it (K <= ring / 2), so the next fragment is
    written while the previous one is in flight.  `records` is an int32 tensor view [ring, E, 2] of that field
    (a CPU tensor in the gloo tests)."""

    def __init__(self, records, world, every=1):
        import torch
        ring = records.shape[0]
        if every < 1 or ring % every or ring // every < 2:
            raise ValueError(f"every={every} must divide the record ring ({ring}) and be <= ring / 2")
        self.world = world
        self.records = records
        self.every = every
        self.ring = ring
        n = records.shape[1]
        self.n = n
        self.out = [torch.empty((world, every, n, 2), dtype=torch.int32, device=records.device) for _ in range(2)]
        self.work = [None, None]
        self._frag = 0

    def launch(self, step):
        """Call after (0-based) step `step`.  When the step closes a fragment of `every` steps, starts
        gathering it and returns the output buffer index; otherwise returns None."""
        import torch.distributed as dist
        if (step + 1) % self.every:
            return None
        k = self._frag & 1
        self._frag += 1
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        s0 = (step + 1 - self.every) % self.ring
        src = self.records[s0:s0 + self.every]          # contiguous: `every` whole slots
        if self.world > 1:
            self.work[k] = dist.all_gather_into_tensor(self.out[k].view(self.world * self.every * self.n, 2),
                                                       src.reshape(self.every * self.n, 2), async_op=True)
        else:
            self.out[k][0].copy_(src)
        return k

    def wait(self, k=None):
        for i in ([k] if k is not None else [0, 1]):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def result(self, k, j=None):
        """Records of buffer k: (reward f32[world*E], status u8[world*E, 4]) of the fragment's j-th step
        (default: its last), rank-major env order."""
        self.wait(k)
        j = self.every - 1 if j is None else j
        return unpack_record(self.out[k][:, j].reshape(self.world * self.n, 2))
