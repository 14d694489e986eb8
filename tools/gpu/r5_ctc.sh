# round 5: CTC kernel time, base vs current library (rocprofv3 kernel stats of a 3-step run)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=$GRAFT_REPO_ROOT/gpurun_out/r5_ctc.log
: > $L
cd /tmp && export TMPDIR=/tmp
for v in base cur; do
  if [ $v = base ]; then export AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_base.so; else unset AVEC_LIB_PATH; fi
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/b_$v.log 2>&1
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v ($f)" >> $L
  if [ -n "$f" ]; then grep -i 'ctc\|softmax' "$f" >> $L; else tail -5 /tmp/b_$v.log >> $L; fi
done
cat $L
