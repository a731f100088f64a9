"""Time the regress launch alone (2000 proposals, both levels) for the modes given on the command line."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
if os.environ.get("ZERO_W"):      # power experiment: all-zero convolution weights (same instruction stream, no operand toggling)
    for k in list(sd):
        if ".conv.0." in k or ".conv.2." in k: sd[k] = sd[k] * 0
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
mid = ops.RegressorWeights(sub("regress_mid."), dev); fine = ops.RegressorWeights(sub("regress_fine."), dev)
H, W, n = 480, 640, int(os.environ.get("NPROP", "2000"))
p1 = synthetic.make_pyramid(7, H, W); p2 = synthetic.make_pyramid(8, H, W)
g = torch.Generator().manual_seed(9)
props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1).to(dev)
g1 = [t.to(dev) for t in p1[:4]]; g2 = [t.to(dev) for t in p2[:4]]
for mode in sys.argv[1:]:
    mid.set_mode(mode); fine.set_mode(mode)
    for _ in range(2): ops.regress(mid, fine, g1, g2, props)
    torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get('NITER', '5'))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.regress(mid, fine, g1, g2, props); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{mode} rot={os.environ.get('P2P_SPLIT_ROT','1')} n={n}: median {sorted(ts)[len(ts)//2]:.3f} ms  min {min(ts):.3f}", flush=True)
