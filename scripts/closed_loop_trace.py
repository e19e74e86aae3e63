#!/usr/bin/env python3
"""A short closed-loop run for rocprofv3 --kernel-trace: G env groups, launcher L, 60 steps at the metric size.
    rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python scripts/closed_loop_trace.py 4 thread"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tactics2d_amd import scenarios as S  # noqa: E402
from tactics2d_amd.debug import ClosedLoop, env_groups as EnvGroups  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
launcher = sys.argv[2] if len(sys.argv) > 2 else "thread"
raw = len(sys.argv) > 3 and sys.argv[3] == "raw"
sc = S.mixed(4096, 64, seed=3)
eg = EnvGroups(sc, G, raw_streams=raw)
eg.configure(lambda p: p.set_auto_reset(True))
loop = ClosedLoop(eg, launcher, 100, graph_steps=20)
loop.run(600)
torch.cuda.synchronize()
loop.run(60)
torch.cuda.synchronize()
loop.close()
eg.close()
