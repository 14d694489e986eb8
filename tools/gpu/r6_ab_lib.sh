# same-box A/B of the step over library builds under tools/_bin (AVEC_LIB_PATH): bash tools/gpu/r6_ab_lib.sh name1 name2 ... ("default" = the shipped library)
mkdir -p gpurun_out/r06; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for rep in 1 2; do for n in "$@"; do
if [ "$n" = "default" ]; then unset AVEC_LIB_PATH; else export AVEC_LIB_PATH=$PWD/tools/_bin/libavec_$n.so; fi
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['value'], d['config']['loss'])"
done; done
