# ablation of conv3x3_shift_kernel (tools/build_abl.sh 1 2 3 4 5 6 7 8 12): bits 1 no MFMA, 2 no LDS-DMA, 4 no fragment reads, 8 no barrier
cd $GRAFT_REPO_ROOT
for n in ${ABLS:-0 1 2 3 4 6 7}; do
  if [ $n = 0 ]; then L=$GRAFT_REPO_ROOT/avec_amd/libavec_hip.so; else L=$GRAFT_REPO_ROOT/tools/_bin/libavec_abl_$n.so; fi
  echo "== ABL $n"
  AVEC_LIB_PATH=$L python tools/abl_gemm.py 2>&1 | grep "conv fwd"
done
