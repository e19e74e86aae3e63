cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0, '.')
import bench, torch
dev = torch.device("cuda", 0)
sc = bench.build_scene("cfg2", *bench.DEFAULTS["cfg2"], seed=0)
r = bench.Runner(sc, dev, "fast")
r.steps_single(2000); torch.cuda.synchronize()
r.pool.profile_enable(True)
r.steps_chain(64, 32); r.steps_single(8)
torch.cuda.synchronize()
for kid in (2, 7): print(kid, r.pool.profile_read(kid))
r.close()
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r16 -o x -- python bench.py --config cfg2 --mode chain --steps 256 --warmup 32 --no-cpu-baseline --no-profile --no-alternates > gpurun_out/r16.log 2>&1
cat gpurun_out/r16/*/x_kernel_stats.csv | cut -c1-200 | head -8
