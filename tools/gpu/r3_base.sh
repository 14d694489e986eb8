mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r3_base.json 2> gpurun_out/r3_base.err
timeout 600 python tools/find_copies.py > gpurun_out/r3_copies.log 2>&1
tail -3 gpurun_out/r3_base.err
head -c 3000 gpurun_out/r3_base.json
