import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
import numpy as np, torch, torch.distributed as dist
from tactics2d_amd import layout as L, scenarios as S, dist as D
from tactics2d_amd.pool import ParticipantPool
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
sc = S.mixed(4096, 64, seed=3); pool = ParticipantPool(sc.n_env, sc.A, 0); sc.load(pool); pool.set_auto_reset(True)
rng = np.random.default_rng(0); a0, a1 = sc.sample_actions(rng)
t0_, t1_ = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
pool.bind_actions(t0_.data_ptr(), t1_.data_ptr())
rec = torch.as_tensor(pool.device_array(L.F_RECORD), device=dev).view(torch.int32)
print("record tensor shape", rec.shape)
stream = torch.cuda.current_stream().cuda_stream
class G(D.ResultGather):
    # world = 1 would short-cut to a copy: force the RCCL call to see its launch / stream-event cost
    def __init__(self, records, every):
        super().__init__(records, 1, every)
        self.world = 2; self._w = 1
    def launch(self, step):
        self.world = 1
        if (step + 1) % self.every: return None
        k = self._frag & 1; self._frag += 1
        if self.work[k] is not None: self.work[k].wait(); self.work[k] = None
        s0 = (step + 1 - self.every) % self.ring
        src = self.records[s0:s0 + self.every]
        self.work[k] = dist.all_gather_into_tensor(self.out[k].view(self.every * self.n, 2), src.reshape(self.every * self.n, 2), async_op=True)
        return k
for mode in ("none", "nccl every 1", "nccl every 8"):
    g = G(rec, int(mode.split()[-1])) if mode != "none" else None
    for k in range(50):
        pool.step(100, stream); 
        if g: g.launch(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 2000
    for k in range(n):
        pool.step(100, stream)
        if g: g.launch(k)
    if g: g.wait()
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(mode, "us/step", 1e6 * el / n)
# host-only cost of the launch calls
t = time.perf_counter()
for k in range(2000): pool.step(100, stream)
print("host enqueue us/step (no sync)", 1e6 * (time.perf_counter() - t) / 2000)
torch.cuda.synchronize()
dist.destroy_process_group()
