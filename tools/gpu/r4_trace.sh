# round 4: per-launch in-graph timeline of a conformer block + prologue / epilogue ablations of the 64x64 product
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_trace.log
: > $L
PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 32 100 256 >> $L 2>&1
PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 1 100 256 >> $L 2>&1
for n in 15 31 47; do
echo "== AVEC_ABL=$n (15: no K loop work; +16: no final stores; +32: no epilogue)" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_abl_$n.so PYTHONPATH=$PWD timeout 300 python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
done
cat $L
