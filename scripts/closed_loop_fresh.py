#!/usr/bin/env python3
"""The closed loop of G env groups in a FRESH process (stream -> hardware-queue mapping depends on what the process created before):
    GPU_MAX_HW_QUEUES=4 python scripts/closed_loop_fresh.py G launcher [raw]
The group streams are created before anything else touches the GPU when `raw` is given (the library's own hipStreamNonBlocking
streams, the first HSA queues of the process)."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
launcher = sys.argv[2] if len(sys.argv) > 2 else "threads"
raw = len(sys.argv) > 3 and sys.argv[3] == "raw"
import torch  # noqa: E402

from tactics2d_amd import scenarios as S  # noqa: E402
from tactics2d_amd.debug import ClosedLoop, env_groups as EnvGroups  # noqa: E402

sc = S.mixed(4096, 64, seed=3)
eg = EnvGroups(sc, G, raw_streams=raw)
eg.configure(lambda p: p.set_auto_reset(True))
loop = ClosedLoop(eg, launcher, 100, graph_steps=64)


def timed(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    loop.run(n)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / n


loop.run(3000)
torch.cuda.synchronize()
us = min(timed(400) for _ in range(3))
loop.run(600)
us20 = min(timed(20) for _ in range(5))
print(json.dumps(dict(q=os.environ["GPU_MAX_HW_QUEUES"], groups=G, launcher=launcher, raw=raw, us_per_step=us, us_per_step_20=us20)), flush=True)
loop.close()
eg.close()
