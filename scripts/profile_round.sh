#!/bin/bash
# Round profile of the headline bench (both step modes): rocprofv3 kernel-trace stats + HBM traffic and SQ counters in
# separate PMC passes (never combined with sys/hip tracing) + the calibration of FETCH_SIZE / WRITE_SIZE, then kernel-trace
# summaries of cfg3 / cfg4 / cfg5 and kernel-trace + SQ passes of the next rows' kernels (ego step, lidar, IDM, scene commit).
# Usage on the GPU box: bash scripts/profile_round.sh r03a     -> gpurun_out/<tag>_*.json / .csv (copy to profiles/)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
COMMON="--steps 1024 --warmup 128 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-closed-loop"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
for MODE in chain step; do
  BENCH="python bench.py --mode $MODE $COMMON"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -o bench -- $BENCH > $OUT/bench_trace_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$MODE -o bench -- $BENCH --no-profile > $OUT/bench_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$MODE -o bench -- $BENCH --no-profile > $OUT/bench_write_$MODE.log 2>&1
  # (the SQ_WAIT_* counters slow the kernel by ~20 %: not collected with the ones ratios are taken from)
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/sq_$MODE -o bench -- $BENCH --no-profile > $OUT/bench_sq_$MODE.log 2>&1
  # class-resolved VALU instruction counts (bench.py roofline.issue_cycles_per_class): two more passes, 8 counters each
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/sqc1_$MODE -o bench -- $BENCH --no-profile > $OUT/bench_sqc1_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/sqc2_$MODE -o bench -- $BENCH --no-profile > $OUT/bench_sqc2_$MODE.log 2>&1
done
# north_star's roofline kernel: an SQ pass of a run whose per-kernel pass launches the stand-alone integrator
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/sq_integ -o bench -- python bench.py --mode step --steps 64 --warmup 32 --clock-warm 0 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-closed-loop > $OUT/bench_sq_integ.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- python scripts/calib_traffic.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- python scripts/calib_traffic.py > $OUT/cal_write.log 2>&1
# the other BASELINE.json configurations (per-GPU shards), kernel time per config
for CFG in cfg2 cfg3 cfg4 cfg5; do
  for MODE in chain step; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${CFG}_$MODE -o bench -- python bench.py --config $CFG --mode $MODE --steps 512 --warmup 64 --no-cpu-baseline --no-profile --no-alternates --no-closed-loop > $OUT/${CFG}_$MODE.log 2>&1
  done
done
# SQ pass of the small pools' t2d_step_n launches (VALU issue per SIMD and step, VALU busy: DESIGN.md 8.16)
for CFG in cfg2 cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/${CFG}_chain_sq -o bench -- python bench.py --config $CFG --mode chain --steps 512 --warmup 64 --no-cpu-baseline --no-profile --no-alternates --no-closed-loop > $OUT/${CFG}_chain_sq.log 2>&1
done
# next rows: kernel-trace, then an SQ pass (VALU busy, instructions per wave) of the same commands
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vec -o vec -- python scripts/time_vec_env.py 4096 > $OUT/vec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/idm -o idm -- python scripts/time_idm.py 4096 > $OUT/idm.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/vec_sq -o vec -- python scripts/time_vec_env.py 4096 short > $OUT/vec_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/idm_sq -o idm -- python scripts/time_idm.py 4096 short > $OUT/idm_sq.log 2>&1
# the Gym-API host path (round 5): kernels of one VecParkingEnv.step at 4096 envs (ego step, lidar, frame pack) and its wall time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/host -o host -- python scripts/time_host_path.py 4096 200 > $OUT/host.log 2>&1
grep '^{' $OUT/host.log > gpurun_out/${TAG}_host_path.json 2>/dev/null
python scripts/summarize_profile.py $OUT $TAG
