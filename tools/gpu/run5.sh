mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/r2_ffn_dbg.log
for d in 0 1 2 3 4 7 8 15 16 31; do
  echo "== dbg $d" >> gpurun_out/r2_ffn_dbg.log
  AVEC_FFN_DBG=$d timeout 300 python tools/bench_ffn.py 2>/dev/null | grep "fused=1" | grep "M=3200 D=256\|M=1600 D=360" >> gpurun_out/r2_ffn_dbg.log
done
