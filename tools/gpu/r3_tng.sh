# grouped weight-gradient launch: XCD-aware workgroup order on / off x reduction rows per tile (profiles/r03_tn_grouped.txt)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for x in "AVEC_NO_XCD_MAP=1" "AVEC_DUMMY=1"; do
for kt in 32 64; do
  echo "== $x KT $kt"; env $x AVEC_TNG_KT=$kt python tools/bench_tn_grouped.py 2>&1 | grep -v amdgpu.ids
done
echo "== $x conv wgrads"; env $x python tools/bench_gemm.py 2>&1 | grep -i "tn plain\|conv wgrad \|conv fwd  3200\|conv wgrad$"
done
timeout 600 python -m pytest tests -q -m gpu -x -k "tn or wgrad or conv2d" 2>&1 | tail -3
