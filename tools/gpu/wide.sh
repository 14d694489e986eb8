mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/wide.log
: > $L
for n in 0 1; do
  AVEC_NT_WIDE=$n timeout 300 python tools/wide_conv_check.py 2>&1 | grep -v "amdgpu.ids" >> $L
done
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resnet or conv or full_model or gemm or visual" 2>&1 | tail -4 >> $L
for n in 0 1; do
AVEC_NT_WIDE=$n python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIDE=$n', d['ms_per_step'], d['value'])" >> $L
done
AVEC_NO_PERM2=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_PERM2', d['ms_per_step'], d['value'])" >> $L
