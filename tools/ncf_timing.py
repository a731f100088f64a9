"""Phase ticks of the fused consensus kernel (needs a -DNCF_TIMING build: P2P_LIB_PATH=tools/exp/lib_nct.so)."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"], sd["ncn.conv.2.bias"], dev)
B = int(os.environ.get("BATCH", "16"))
x = torch.rand(B, 30, 40, 30, 40, device=dev)
for _ in range(3):
    y = ops.neigh_consensus_batch(x, ncn)
torch.cuda.synchronize()
d = y[0].reshape(-1)[:40].cpu()
names = ["prologue (full stage)", "S2 layer 1 (+ ring loads)", "barrier 1", "S3 layer 2", "S1 stores of the next strip", "flush (top of strip)", "barrier 2", "-"]
n = float(d[32])
print(f"ticks of work-group 5 (direct branch, pair 0), {n:.0f} strips, batch {B}: total per wave / per strip")
for w in range(4):
    v = d[w * 8:w * 8 + 8]
    print(f"  wave {w}: " + ", ".join(f"{nm} {float(t):.0f} / {float(t) / n:.0f}" for nm, t in zip(names[:7], v[:7])) + f"; sum {float(v[:7].sum()):.0f} / {float(v[1:7].sum()) / n:.0f}")
