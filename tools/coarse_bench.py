"""Time the coarse stage alone (p2p_coarse_forward + p2p_coarse_matches, one 480x640 pair, ksize 2)."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"], sd["ncn.conv.2.bias"], dev)
if os.environ.get("TILE"):      # force the consensus kernel's work-group tile "ta,tb,tc"
    ncn.set_tile(*[int(v) for v in os.environ["TILE"].split(",")])
H, W = int(os.environ.get("H", "480")), int(os.environ.get("W", "640"))
p1, p2 = synthetic.make_correlated_pyramids(3, H, W)
B = int(os.environ.get("BATCH", "1"))
fa, fb = p1[4].to(dev)[None].repeat(B, 1, 1, 1), p2[4].to(dev)[None].repeat(B, 1, 1, 1)
reps = int(os.environ.get("REPS", "50"))
for _ in range(5): ops.coarse_forward_batch(fa, fb, 2, ncn)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): ops.coarse_forward_batch(fa, fb, 2, ncn)
b.record(); torch.cuda.synchronize()
print(f"coarse_forward {H}x{W} batch {B}: {a.elapsed_time(b) / reps / B * 1e3:.1f} us per pair  (lib {os.environ.get('P2P_LIB_PATH', 'default')}, tile {os.environ.get('TILE', 'auto')})", flush=True)
