#!/bin/bash
# The N = 8 (and N = 4) bench line rehearsed on ONE GPU: eight ranks over gloo, every rank on device 0 (T2D_FORCE_DEVICE) -- the
# driver's own command line otherwise.  Checks the shard sizes, the gather cadence and the fixed-total (strong) cases of the
# multi-GPU run end to end; the RCCL communicator itself needs eight GPUs.  Numbers mean nothing (eight ranks share one GPU).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export T2D_DIST_BACKEND=gloo T2D_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
      bench.py --gpus $n --steps 20 --warmup 5 > gpurun_out/rehearse_n$n.json 2> gpurun_out/rehearse_n$n.err
  echo "N=$n rc $?"
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
lines = [l for l in open(f"gpurun_out/rehearse_n{n}.json").read().strip().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d[k] for k in ("n_gpus", "steps", "scaling", "value", "ms_per_step")})
print("gather", {k: d["gather"][k] for k in ("native", "every", "gathers_in_timed_region") if k in d["gather"]} if d.get("gather") else None)
for k, v in (d.get("strong") or {}).items():
    if not isinstance(v, dict):
        continue
    print("strong", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("envs_per_rank", "us_per_step", "value", "form", "skipped", "gather_every", "kernel_form")})
PY
  tail -n 3 gpurun_out/rehearse_n$n.err
done
