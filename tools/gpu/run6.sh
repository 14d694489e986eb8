mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/r2_small_gemm.log
for cfg in "AVEC_X=0" "AVEC_NT_TILE=64" "AVEC_NT_TILE=12864" "AVEC_NT_TILE=128" "AVEC_NO_GLDS=1 AVEC_NT_TILE=64" "AVEC_NO_GLDS=1 AVEC_NT_TILE=12864" "AVEC_NO_GLDS=1 AVEC_NT_TILE=128" "AVEC_NT_TILE=64 AVEC_NT_RB=64" "AVEC_NT_TILE=128 AVEC_NT_RB=64"; do
  echo "== $cfg" >> gpurun_out/r2_small_gemm.log
  env $cfg timeout 300 python tools/bench_small_gemm.py 2>/dev/null >> gpurun_out/r2_small_gemm.log
done
