# round 5, experiment 1: baseline step on this box + what the shifted-window convolution's time is made of (ablation builds: 7 no MFMA/DMA/fragment reads, 16 no global
# stores, 32 no epilogue, 39 = 7 + 32) + channel sweep (per-tile fixed cost)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r5_e1.log
: > $L
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r5_e1.err | grep "^{" > gpurun_out/r5_e1_bench.json
python -c "import json; d=json.loads(open('gpurun_out/r5_e1_bench.json').read().strip().splitlines()[-1]); print('baseline', d['ms_per_step'], d['value'])" >> $L
for n in 0 7 16 32 39; do
  if [ $n = 0 ]; then LIBP=$PWD/avec_amd/libavec_hip.so; else LIBP=$PWD/tools/_bin/libavec_abl_$n.so; fi
  echo "== ABL $n" >> $L
  if [ $n = 0 ]; then AVEC_LIB_PATH=$LIBP python tools/abl_conv.py --sweep 2>&1 | grep "conv" >> $L; else AVEC_LIB_PATH=$LIBP python tools/abl_conv.py 2>&1 | grep "conv" >> $L; fi
done
echo "== small gemm" >> $L
python tools/bench_small_gemm.py 2>&1 | tail -19 >> $L
cat $L
