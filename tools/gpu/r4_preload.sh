# round 4: kernel-argument preload (-mllvm -amdgpu-kernarg-preload-count=16): parity subset, per-launch block trace, same-box A/B of the step against the HEAD library
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_preload.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_layers.py -m gpu -q 2>&1 | tail -2 >> $L
echo "== block trace, HEAD library" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 32 100 256 2>&1 | grep -v amdgpu >> $L
echo "== block trace, kernel-argument preload" >> $L
PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 32 100 256 2>&1 | grep -v amdgpu >> $L
for rep in 1 2 3; do
for cfg in "AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so" "AVEC_X=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_preload.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
