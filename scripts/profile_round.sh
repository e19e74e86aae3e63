#!/bin/bash
# Round profile: kernel-trace stats of the default bench + HBM traffic counters in separate PMC
# passes (never combined with sys/hip tracing).  Usage on the GPU box: bash scripts/profile_round.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# --groups 1 = the headline configuration only (the default run appends the 4-group pipelined measurement, whose
# quarter-size launches of the same kernel would otherwise be averaged into the same kernel-stats row)
BENCH="python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-configs --groups 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $BENCH --no-profile > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $BENCH --no-profile > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- python scripts/calib_traffic.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- python scripts/calib_traffic.py > $OUT/cal_write.log 2>&1
# (the SQ_WAIT_* counters slow the kernel by ~20 %: not collected with the ones ratios are taken from)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o bench -- $BENCH --no-profile > $OUT/bench_sq.log 2>&1
python scripts/summarize_profile.py $OUT $TAG
