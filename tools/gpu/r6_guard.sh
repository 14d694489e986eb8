cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; export PYTHONPATH=$PWD OMP_NUM_THREADS=4 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in tail head; do for cfg in "--batch 2 --dtype bf16" "--batch 1 --dtype bf16 --dist" "--batch 3 --dtype f32" "--batch 1 --dtype f32 --dist" "--batch 5 --dtype bf16 --dist" "--batch 7 --dtype bf16"; do
  echo "== $mode $cfg"; GUARD_MODE=$mode timeout 900 python -X faulthandler tools/guard/guard_pass.py $cfg 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp\|^  File\|Extension modules" | tail -4 | cut -c1-300
done; done
