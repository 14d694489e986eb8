# gathered weight-gradient kernel: split target (AVEC_TN_WGS) on the three stride-2 layers
cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for w in 256 384 512 640 768 896; do echo "== AVEC_TN_WGS=$w"; AVEC_TN_WGS=$w python tools/bench_gemm.py 2>&1 | grep -A4 " s2 " | grep "conv wgrad *[0-9]"; done
