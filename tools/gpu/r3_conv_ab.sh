# usage: bash tools/gpu/r3_conv_ab.sh "nameA nameB"  -- conv micro-benchmarks (tools/bench_gemm.py) with variant libraries tools/_bin/libavec_<name>.so
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for v in $1; do
echo "== $v"
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_$v.so python tools/bench_gemm.py bf16 2>&1 | grep -E "conv (fwd|bwd)" 
done
