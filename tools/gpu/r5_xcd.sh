mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tools
L=gpurun_out/r5_xcd.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round5.py -x -q -m gpu -k "conv or resnet or shift" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 >> $L
for v in 1 0; do
AVEC_SHIFT_XCD=$v python - <<'P' 2>&1 | grep -v amdgpu | grep "conv fwd\|bwd-data" | sed "s/^/XCD=$v /" >> $L
import sys; sys.argv=['x','bf16']
import builtins
import bench_gemm as b
P
done
for rep in 1 2; do for v in 0 1; do
AVEC_SHIFT_XCD=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD=$v STEP', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done; done
cat $L
