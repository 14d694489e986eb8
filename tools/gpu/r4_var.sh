mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_var.log
: > $L
timeout 900 python tools/bench_variants.py --only lrs2_main 2>>gpurun_out/r4_var.err | grep "^{" >> $L
timeout 900 python tools/bench_variants.py --only lrs2_main_bucketed 2>>gpurun_out/r4_var.err | grep "^{" >> $L
timeout 900 python tools/bench_variants.py --only lrs2_main_bucketed --graphs 2>>gpurun_out/r4_var.err | grep "^{" >> $L
cat $L; grep -v "amdgpu.ids" gpurun_out/r4_var.err | tail -5
