mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/dge.log
: > $L
i=0
for mode in eager eager graph graph eager graph; do
  i=$((i+1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) tools/ddp_graph_equiv.py --out /tmp/dge_$i.pt --mode $mode > /dev/null 2>&1
done
python - >> $L <<'PY'
import torch, itertools
modes = ["eager","eager","graph","graph","eager","graph"]
r = [torch.load("/tmp/dge_%d.pt" % (i+1)) for i in range(6)]
for i, j in itertools.combinations(range(6), 2):
    a, b = r[i]["exp_avg"].double(), r[j]["exp_avg"].double()
    print(modes[i], i, modes[j], j, "%.3e" % ((a-b).norm()/a.norm()).item())
PY
