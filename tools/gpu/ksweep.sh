mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
: > gpurun_out/ksweep.log
for rb in 64 128; do AVEC_NT_RB=$rb python tools/ksweep.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ksweep.log; done
