cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for w in 64 128 192 256 320 384 512; do echo "== AVEC_TNG_WGS=$w"; AVEC_TNG_WGS=$w python tools/bench_tn_grouped.py 2>&1 | grep -v "amdgpu\|work-list"; done
bash tools/gpu/r3_ab4.sh 2 "AVEC_TNG_WGS=384" "AVEC_TNG_WGS=256" "AVEC_TNG_WGS=192" "AVEC_TNG_WGS=512" > /dev/null 2>&1; cat gpurun_out/r3_ab.log
