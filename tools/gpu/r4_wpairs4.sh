mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_wpairs4.log
: > $L
for n in 3200 800 200; do
for cfg in "AVEC_X=0" "AVEC_LIB_PATH=tools/_bin/libavec_wp_abl_1.so"; do
  echo "== images $n $cfg" >> $L
  env WG_IMAGES=$n $cfg PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids | grep grouped >> $L
done
done
cat $L
