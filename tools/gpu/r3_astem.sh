cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
echo "== first version (AVEC_NO_ASTEM_X=1)"; AVEC_NO_ASTEM_X=1 python tools/bench_audio_stem.py 2>&1 | grep -v amdgpu
for l in "" $(ls tools/_bin/libavec_as*.so 2>/dev/null); do echo "== ${l:-default}"; AVEC_LIB_PATH=${l:+$GRAFT_REPO_ROOT/$l} python tools/bench_audio_stem.py 2>&1 | grep -v amdgpu; done
timeout 900 python -m pytest tests -q -x -m gpu -k "audio or stem or golden or full_model" 2>&1 | tail -3
