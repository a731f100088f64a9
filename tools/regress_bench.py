"""Time the regress launch alone (both levels) for the modes given on the command line.
NPROP proposals in total (default 2000), spread over NPAIRS image pairs with their own pyramids (default 1; the benched
configuration is NPAIRS=16 NPROP=6400: 16 x 61 MB of pyramids do not stay in the caches, a single pair's do)."""
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
H, W, n, npairs = 480, 640, int(os.environ.get("NPROP", "2000")), int(os.environ.get("NPAIRS", "1"))
g = torch.Generator().manual_seed(9)
per = n // npairs
pyr1, pyr2, props = [], [], []
for i in range(npairs):
    p1 = synthetic.make_pyramid(7 + 2 * i, H, W); p2 = synthetic.make_pyramid(8 + 2 * i, H, W)
    pyr1.append([t.to(dev) for t in p1[:4]]); pyr2.append([t.to(dev) for t in p2[:4]])
    props.append(torch.stack([torch.randint(0, W + 1, (per,), generator=g), torch.randint(0, H + 1, (per,), generator=g),
                              torch.randint(0, W + 1, (per,), generator=g), torch.randint(0, H + 1, (per,), generator=g)], 1).to(dev))
run = lambda: ops.regress_batch(mid, fine, pyr1, pyr2, props)
for mode in sys.argv[1:]:
    mid.set_mode(mode); fine.set_mode(mode)
    for _ in range(2): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get('NITER', '5'))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{mode} n={per}x{npairs}: median {sorted(ts)[len(ts)//2]:.3f} ms  min {min(ts):.3f}", flush=True)
