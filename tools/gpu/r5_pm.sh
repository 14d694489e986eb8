mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tools
python - <<'P' 2>&1 | grep -v amdgpu > gpurun_out/r5_pm.log
import sys; sys.argv=['x','bf16']
import bench_gemm as b
from avec_amd.lib import lib
for (M,N,K) in [(3200,4608,4608),(3200,4608,2560),(3200,512,4608),(3200,2304,2304)]:
    b.plain(M,N,K); print("   last kernel:", lib.raw("avec_last_kernel")().decode())
b.conv(3200,3,512,512); b.conv(3200,6,256,256)
P
cat gpurun_out/r5_pm.log
