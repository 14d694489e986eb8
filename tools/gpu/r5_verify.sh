# round 5 re-entry: full GPU suite + smoke + quick bench on the restored tree
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -15 > gpurun_out/r5_verify_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5_verify_smoke.log 2>&1
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>gpurun_out/r5_verify_bench.err | grep "^{" > gpurun_out/r5_verify_bench.json
tail -3 gpurun_out/r5_verify_tests.log; tail -1 gpurun_out/r5_verify_smoke.log
python -c "import json; d=json.loads(open('gpurun_out/r5_verify_bench.json').read().strip().splitlines()[-1]); print('STEP', d['ms_per_step'], d['value'])"
