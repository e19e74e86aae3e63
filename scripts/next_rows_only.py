#!/usr/bin/env python3
"""bench.py's `next_rows` alone (IDM, lidar, the ParkingEnv vector step), same scenes, same clock ramp, one JSON line:

    python scripts/next_rows_only.py"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
scene = bench.build_scene("metric", *bench.DEFAULTS["metric"], seed=0)
warm = bench.Runner(scene, dev, "fast", auto_reset=True, outputs="all", seed=7)


def clock_warm():
    warm.steps_single(3000)
    torch.cuda.synchronize()


rows = bench.next_rows(dev, clock_warm, scene)
print(json.dumps({k: v for k, v in rows.items() if not k.endswith("note")}))
