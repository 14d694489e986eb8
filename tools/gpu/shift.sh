mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/shift.log
: > $L
AVEC_NO_CONV_SHIFT=1 timeout 300 python tools/wide_conv_check.py 2>&1 | grep -v "amdgpu.ids" | grep "s1" >> $L
timeout 300 python tools/wide_conv_check.py 2>&1 | grep -v "amdgpu.ids" | grep "s1" >> $L
