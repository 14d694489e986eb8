mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${KEXPR:-gemm or conv or linear or resnet}" > gpurun_out/r3_q_test.log 2>&1
tail -4 gpurun_out/r3_q_test.log
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss'])"; done
