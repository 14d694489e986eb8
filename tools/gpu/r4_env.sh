# round 4: runtime knobs of the HIP runtime that touch kernel-argument placement / graph packet building / hardware queues: same-box A/B of the step (env only)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_env.log
: > $L
env | grep -i "HIP_\|HSA_\|GPU_MAX\|DEBUG_CLR" >> $L
for rep in 1 2; do
for cfg in "AVEC_X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2"; do
env $cfg timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_env.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
