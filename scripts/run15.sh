cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_ego.py tests/test_iou_events.py tests/test_gpu_envs.py -m gpu -q -x > gpurun_out/r15_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r15_tests.log
tail -5 gpurun_out/r15_tests.log
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench, torch
dev = torch.device("cuda", 0)
def cw():
    pass
for name in ("cfg2",):
    sc = bench.build_scene(name, *bench.DEFAULTS[name], seed=0)
    r = bench.Runner(sc, dev, "fast")
    r.steps_single(3000); torch.cuda.synchronize()
    for mode in ("step", "chain", "step", "chain"):
        us, span = bench.timed(r, mode, 2048, 128, 32)
        print(name, mode, round(us, 2), round(span, 2))
    r.close()
PY
