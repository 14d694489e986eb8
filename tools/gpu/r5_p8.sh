# round 5: why does the 256x256 / BK 64 / 8-wave GEMM microbenchmark stop at 950-1000 TFLOP/s: ablation variants + s_memtime stamps of every phase
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r5_p8.log
: > $L
for t in 0 2; do
echo "== P8_TRACE=$t" >> $L
timeout 300 tools/_bin/ubench_gemm_p8_t$t >> $L 2>&1
done
cat $L
