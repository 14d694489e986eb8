mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_slab_abl.log
: > $L
echo "== full" >> $L
python tools/bench_slab.py 2>&1 | grep slab >> $L
for n in 1 2 4 8 9 13 15; do
echo "== C3S_ABL $n" >> $L
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_c3sabl_$n.so timeout 300 python tools/bench_slab.py 2>&1 | grep "slab fwd\|slab bwd" >> $L
done
cat $L
