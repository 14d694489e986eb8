mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_bisect.log
: > $L
for lib in tools/_bin/libavec_base.so tools/_bin/libavec_batched.so tools/_bin/libavec_head.so avec_amd/libavec_hip.so; do
echo "== $lib" >> $L
AVEC_LIB_PATH=$PWD/$lib timeout 600 python -m pytest "tests/test_gpu_round3.py::test_resnet_block_relu_bitmask_equals_reading_the_saved_output" -m gpu -x -q 2>&1 | grep "passed\|failed\|^E   " | head -4 >> $L
done
cat $L
