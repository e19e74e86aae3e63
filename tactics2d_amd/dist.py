"""Multi-GPU plumbing: one process per GPU, environments sharded in contiguous blocks, and the
only collective on the path -- an all-gather of the 8-byte per-env result record {reward f32,
scenario_status u8, traffic_status u8, terminated u8, truncated u8}.

Environments never interact (traffic/scenario_manager.py:52-61: one ScenarioManager owns one
scene), so there is no data-path exchange: every rank steps its own pool.  On GPUs the collective
is issued by the library itself (`NativeGather` -> t2d_gather: RCCL all-gather over xGMI reading
the pool's record ring in place, on a stream of the pool's own); torch.distributed is the
BOOTSTRAP only (rendezvous, shipping the 128-byte communicator id, the bench's barrier).
`ResultGather` is the same exchange through torch.distributed, kept for backends without RCCL
(the world_size-2 gloo tests on CPU and on a one-GPU box).
"""
import os



def env_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults 0, 0, 1)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(n_env_total, rank, world, allow_uneven=False):
    """Contiguous block sharding: rank r owns envs [r*E/G, (r+1)*E/G).  The result gather ships equally sized
    fragments (all_gather requires it, and result() assumes a rank-major layout), so E must be a multiple of G;
    allow_uneven=True gives the low ranks the remainder for jobs that never gather."""
    base, rem = divmod(n_env_total, world)
    if rem and not allow_uneven:
        raise ValueError(f"{n_env_total} environments do not split evenly over {world} ranks "
                         f"(the result gather needs equal shards; pad the job or pass allow_uneven=True)")
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
    K must divide the ring and leave at least two fragments in it (K <= ring / 2).  At most ONE gather is in flight:
    launch() first waits (a stream wait for RCCL, a host wait for gloo) for the previous fragment's gather -- the steps
    the caller enqueues next are the ones that overwrite the slots that gather was reading, and the previous output
    buffer is free again.  `records` is an int32 tensor view [ring, E, 2] of that field (a CPU tensor in the gloo
    tests)."""

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
        # every earlier gather must be done before the caller's next steps reuse its ring slots (with two fragments in
        # the ring the very next step does), and before its output buffer is written again
        self.wait()
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


class NativeGather:
    """The result gather issued by the library (t2d_gather): an RCCL all-gather over xGMI that reads the pool's record
    ring in place, on a stream owned by the pool, ordered after the steps' stream by events -- the steps that follow a
    fragment do not wait for it, and a step that would overwrite a slot still being read waits inside t2d_step.
    torch is used for the two output tensors only.  Same calling convention as ResultGather."""

    def __init__(self, pool, world, every=16, device=None):
        import torch
        from . import layout as L
        if every < 1 or L.RECORD_RING % every or L.RECORD_RING // every < 2:
            raise ValueError(f"every={every} must divide the record ring ({L.RECORD_RING}) and be <= ring / 2")
        self.pool, self.world, self.every, self.n = pool, world, every, pool.n_env
        self.out = [torch.empty((world, every, self.n, 2), dtype=torch.int32, device=device) for _ in range(2)]
        self._frag = 0
        self._stream = None

    @staticmethod
    def bootstrap(pool, rank, world):
        """Create the pool's RCCL communicator: rank 0 draws the id, torch.distributed (any backend) ships it."""
        uid = [None]
        if world > 1:
            import torch.distributed as dist
            if rank == 0:
                uid[0] = pool.comm_unique_id()
            dist.broadcast_object_list(uid, src=0)
        pool.comm_init(uid[0], rank, world)

    def launch(self, step=None, stream=None):
        """Call after every step (or t2d_step_n fragment) with the stream the steps run on (raw handle or None): a gather is
        issued whenever the POOL's step count (t2d_step_count -- not the caller's index: steps taken before this object
        existed count too) reaches a multiple of `every`.  Returns the output buffer (0 / 1) the fragment goes to, else None.
        Buffer k is written again two fragments later, ordered after whatever `stream` holds at that launch: consume
        result(k) on the steps' stream, or wait(), before launching two more fragments."""
        if self.pool.step_count() % self.every:
            return None
        k = self._frag & 1
        self._frag += 1
        self._stream = stream
        self.pool.gather(self.every, self.out[k].data_ptr(), stream)
        return k

    def wait(self, k=None):
        """Blocks until every gather issued so far has landed (gathers run in order on one stream: buffer k's is among them)."""
        self.pool.gather_wait(self._stream, block_host=True)

    def result(self, k, j=None):
        self.wait(k)
        j = self.every - 1 if j is None else j
        return unpack_record(self.out[k][:, j].reshape(self.world * self.n, 2))
