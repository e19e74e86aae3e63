cd $GRAFT_REPO_ROOT
for K in hw rb ix; do
  T2D_COUNT_CONFIG=$K T2D_COUNT_STEPS=100 bash scripts/sq_variants.sh libt2d_hip.so libt2d_skip1.so libt2d_skip4.so libt2d_skip8.so libt2d_skip16.so
done 2>&1 | grep "^libt2d" | tee gpurun_out/r6_sq_kinds.txt
