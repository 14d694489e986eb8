cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round3.py -q -x -k "depthwise" 2>&1 | tail -4
timeout 900 python -m pytest tests -q -x -m gpu -k "golden or conformer or block or full_model" 2>&1 | tail -2
bash tools/gpu/r3_ab4.sh 3 "AVEC_DUMMY=1" "AVEC_NO_DWCONV_FUSED_BWD=1" > /dev/null 2>&1; cat gpurun_out/r3_ab.log
