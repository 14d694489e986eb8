# round 4: the data-parallel step over a ONE-rank RCCL group (every collective path, in-graph all-reduce) + the two-rank gloo tests + bench under the one-rank group
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_rccl.log
: > $L
timeout 2400 python -m pytest tests/test_gpu_ddp.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -30 >> $L
for rep in 1 2; do
for cfg in "AVEC_DIST_SINGLE=0" "AVEC_DIST_SINGLE=1" "AVEC_DIST_SINGLE=1 AVEC_PEER_FUSED=0" "AVEC_DIST_SINGLE=1 AVEC_PEER_SYNCBN=0"; do
env $cfg timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_rccl.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'], d['config']['step_path'], '| fallback:', d['config']['fallback'], '| syncbn:', d['config']['syncbn_exchange'])" >> $L
done
done
tail -5 gpurun_out/r4_rccl.err >> $L
cat $L
