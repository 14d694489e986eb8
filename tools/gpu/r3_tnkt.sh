cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for kt in 32 64; do for w in 0 1024 2048 4096; do echo "== AVEC_TN_KT=$kt AVEC_TN_WGS=$w"; AVEC_TN_KT=$kt AVEC_TN_WGS=$w python tools/bench_gemm.py 2>&1 | grep -A4 " s2 " | grep "s2\|conv wgrad *[0-9]"; done; done
