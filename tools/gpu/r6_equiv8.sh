# the eight-ranks-on-one-GPU equivalence case of tests/test_gpu_ddp.py, N times: per-rank local losses of every run (AVEC_DDP_DEBUG) next to the single-process loss
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 MASTER_ADDR=127.0.0.1 AVEC_PEER_SYNCBN=${PEER:-1} AVEC_DDP_DEBUG=1
N=${1:-6}; W=${2:-8}; B=${3:-8}
python tools/ddp_equiv.py --out /tmp/single.pt --batch $B 2>/dev/null | grep "local loss"
for i in $(seq 1 $N); do
  AVEC_DDP_DEBUG_DIR=/tmp/dbg_$i timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700+i)) tools/ddp_equiv.py --out /tmp/ddp_$i.pt --backend gloo --batch $B --share-gpu > /tmp/o_$i.txt 2> /tmp/e_$i.txt
  rc=$?
  echo "run $i rc=$rc"; [ -n "$VERBOSE" ] && grep "local loss" /tmp/o_$i.txt | sort -k2n | awk '{printf "%s:%s/%s  ", $2, $5, $7} END {print ""}'
  grep "exchange #" /tmp/o_$i.txt | sort -k5n | head -12
  python - <<PY
import torch
a, b = torch.load("/tmp/single.pt"), torch.load("/tmp/ddp_$i.pt")
d = (a["grad"] - b["grad"]).abs().max().item()
print("   loss single %.7f ddp %.7f  max|dgrad| %.3e  max|grad| %.3e" % (float(a["loss"]), float(b["loss"]), d, a["grad"].abs().max().item()))
PY
  if [ $i -gt 1 ]; then python tools/gpu/r6_equiv8_cmp.py /tmp/dbg_1 /tmp/dbg_$i $W | grep -v ": 0 of" | cut -c1-400; fi
done
