cd $GRAFT_REPO_ROOT
export T2D_DIST_BACKEND=gloo T2D_FORCE_DEVICE=0
for N in 2 8; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 32 --warmup 16 --envs 512 > gpurun_out/r12_n$N.json 2> gpurun_out/r12_n$N.err; echo "N=$N rc=$?"; tail -2 gpurun_out/r12_n$N.err
python - <<PY
import json
d=json.load(open('gpurun_out/r12_n$N.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d['gather'], d['config']['parallelism'])
PY
done
unset T2D_DIST_BACKEND T2D_FORCE_DEVICE
T2D_FORCE_GATHER=1 timeout 600 python bench.py --steps 64 --warmup 16 --gather-every 1 --no-cpu-baseline --no-configs --no-next-rows > gpurun_out/r12_g1.json 2> gpurun_out/r12_g1.err; echo rc=$?
T2D_FORCE_GATHER=1 timeout 600 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-configs --no-next-rows > gpurun_out/r12_g16.json 2> gpurun_out/r12_g16.err; echo rc=$?
python - <<PY
import json
for f in ('r12_g1','r12_g16'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f,{k:d[k] for k in ('value','ms_per_step')}, d['gather'])
PY
python __graft_entry__.py smoke 2>&1 | tail -2
