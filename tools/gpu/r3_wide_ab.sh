# usage: bash tools/gpu/r3_wide_ab.sh "nameA nameB"  -- wgrad micro-benchmarks (tools/bench_gemm.py) with variant libraries
cd $GRAFT_REPO_ROOT
for v in $1; do
echo "== $v"
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_$v.so python tools/bench_gemm.py bf16 2>&1 | grep -E "conv fwd  3200|wgrad" 
done
