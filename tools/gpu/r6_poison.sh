cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; export PYTHONPATH=$PWD OMP_NUM_THREADS=4 HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "$@"; do
  echo "== $cfg"; timeout 600 python tools/poison_check.py $cfg 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp\|UserWarning\|Consider using\|loss, grad" | tail -12 | cut -c1-400
done
