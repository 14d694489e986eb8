cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "stem and not audio" 2>&1 | tail -3
for r in 1 2; do
AVEC_S3P_RING=0 python tools/bench_stem_abl.py 2>&1 | grep ABL | sed 's/^/band  /'
python tools/bench_stem_abl.py 2>&1 | grep ABL | sed 's/^/ring  /'
done
