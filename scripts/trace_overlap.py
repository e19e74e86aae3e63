#!/usr/bin/env python3
"""Read a rocprofv3 kernel trace of a closed-loop run: which queues the step kernels of the last 40 steps ran on, how long
they took, and how much of their time they overlapped another step kernel.
    python scripts/trace_overlap.py OUT/run_kernel_trace.csv"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ks = [r for r in rows if "collide_kernel" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = ks[-160:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in ks]
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
dur = sorted(e - s for s, e, _, _ in iv)
busy = sum(e - s for s, e, _, _ in iv)
# union length of the intervals
u, cur_s, cur_e = 0, None, None
for s, e, _, _ in sorted(iv):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            u += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
pol = [r for r in rows if "feedback_policy" in r["Kernel_Name"]]
pd = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in pol[-160:])
print(json.dumps(dict(step_kernels=len(iv), span_us=(t1 - t0) / 1e3, sum_of_durations_us=busy / 1e3, union_us=u / 1e3,
                      mean_concurrency=busy / max(u, 1), median_step_kernel_us=dur[len(dur) // 2] / 1e3,
                      queues=sorted({q for _, _, q, _ in iv}), streams=sorted({s for _, _, _, s in iv}),
                      median_policy_kernel_us=(pd[len(pd) // 2] / 1e3 if pd else None),
                      columns=list(rows[0].keys()))))
