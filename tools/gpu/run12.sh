mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/r2_bench12.log
run() { echo "== $*" >> gpurun_out/r2_bench12.log; env "$@" 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('hipgraph'), d['config'].get('syncbn_exchange'))" >> gpurun_out/r2_bench12.log; }
run AVEC_X=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run AVEC_X=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --eager
run AVEC_DIAG_SKIP_GRAD_ALLREDUCE=1 python bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
run AVEC_DIAG_SKIP_GRAD_ALLREDUCE=1 python bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --eager
run AVEC_DIAG_SKIP_GRAD_ALLREDUCE=1 AVEC_PEER_SYNCBN=0 python bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
run AVEC_X=1 python bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing
